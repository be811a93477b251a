"""Dev tool: latency of the batched Poseidon transcript kernel (host buffers in/out) for a proof-shaped schedule."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bn254 as O, transcript as T
import snark_verifier_amd as sv
ctx = sv.Context(0)
spec = sv.PoseidonSpec(ctx, 5, 4, 8, 60, T.poseidon_opt_tables(5, 8, 60))
seg = [9, 0, 0, 6, 0, 6, 19, 8]
L = sum(seg); rng = random.Random(1)
for n in (1, 64, 1024, 8192, 32768):
    elems = os.urandom(31 * 1)  # placeholder
    row = b"".join(rng.randrange(O.R).to_bytes(32, "little") for _ in range(L))
    elems = row * n
    for _ in range(2): ctx.poseidon_transcript_batch(spec, elems, n, seg)
    t0 = time.perf_counter()
    for _ in range(3): out = ctx.poseidon_transcript_batch(spec, elems, n, seg)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("n=%d: %.2f ms per batch, %.1f us per transcript" % (n, ms, ms * 1e3 / n))
