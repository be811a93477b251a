"""Time of the batched Poseidon transcript launch (csrc/poseidon.hip) on the shape of a StandardPlonk + GWC19 proof:
n transcripts x 48 elements in 7 segments, host pointers (uploads of 32 n L bytes included).  python tools/poseidon_time.py [n]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import snark_verifier_amd as sv
import transcript as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = sv.Context(0)
spec = sv.PoseidonSpec(ctx, 5, 4, 8, 60, T.poseidon_opt_tables(5, 8, 60))
seg = [9, 0, 0, 6, 0, 6, 19, 8]
L = sum(seg)
rng = random.Random(1)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
elems = b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(n * L))
for _ in range(3):
    out = ctx.poseidon_transcript_batch(spec, elems, n, seg)
best = None
for _ in range(10):
    t0 = time.perf_counter()
    out = ctx.poseidon_transcript_batch(spec, elems, n, seg)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
row = [int.from_bytes(elems[32 * k:32 * k + 32], "little") for k in range(L)]
exp = T.poseidon_transcript_challenges(row, seg)
ok = all(int.from_bytes(out[32 * q:32 * q + 32], "little") == e for q, e in enumerate(exp))
print("poseidon_transcript_batch: %d transcripts x %d elements, %d squeezes: %.3f ms (best of 10), transcript 0 == oracle: %s" % (n, L, len(seg), best * 1e3, ok))
