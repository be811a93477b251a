"""Dev tool (GPU box): hunts a stale-scratch dependence of the tiny-MSM path (n = 1 .. 6: the per-rank step of an n < world
sharding) -- the other candidate cause of the round-5 red test besides the stream race.  One context; every round first
dirties the context's scratch with a large MSM of random size / window, then runs the 8-rank emulation of
tests/test_gpu_robustness.py::test_empty_shard_contributes_the_identity on fresh random inputs and compares the fold with
the oracle.  Also runs the tiny MSMs under every window size the rule can pick, and with the inputs at unaligned-but-legal
offsets inside a poisoned buffer (bytes beyond the shard must never matter).
   python tools/stress_tiny_after_big.py [rounds] > profiles/r06_stress_tiny_after_big.txt"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import coracle as C  # noqa: E402
import snark_verifier_amd as sv  # noqa: E402
from snark_verifier_amd.distributed import gpu_msm_partial, shard_range  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(0x606)
ctx = sv.Context(0)  # private stream, ordered by events (the round-6 product wiring)
big_s = torch.empty(32 << 18, dtype=torch.uint8, device="cuda")
big_p = torch.empty(64 << 18, dtype=torch.uint8, device="cuda")
big_o = torch.zeros(64, dtype=torch.uint8, device="cuda")
ctx.sample_scalars_dev(11, 1 << 18, big_s.data_ptr())
ctx.sample_points_dev(12, 1 << 18, big_p.data_ptr())
ctx.sync()
bad = 0
for it in range(rounds):
    nb = rng.choice([1 << 18, 1 << 17, 77777, 4097, 1 << 14, 33])
    ctx.msm_pippenger_dev(big_s.data_ptr(), big_p.data_ptr(), nb, big_o.data_ptr(), rng.choice([0, 0, 8, 12, 16]))
    n, world = rng.randrange(1, 7), 8
    s, p = C.sample_scalars(1000 + it, n), C.sample_points(5000 + it, n)
    # the shard sits inside a poisoned buffer at a 64-byte-aligned offset: whatever lies beyond it must not matter
    pad = 64 * rng.randrange(0, 4)
    hs = bytearray(os.urandom(pad // 2)) + bytearray(s) + bytearray(b"\xff" * 4096)
    hp = bytearray(os.urandom(pad)) + bytearray(p) + bytearray(b"\xff" * 4096)
    ds = torch.frombuffer(hs, dtype=torch.uint8).cuda()[pad // 2:]
    dp = torch.frombuffer(hp, dtype=torch.uint8).cuda()[pad:]
    gathered = torch.zeros(world, sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        part = torch.full((sv.G1_PARTIAL_BYTES,), 0xAB, dtype=torch.uint8, device="cuda")
        gpu_msm_partial(ctx, part, ds[32 * lo:], dp[64 * lo:], hi - lo, rng.choice([0, 0, 0, 2, 3, 5, 8]))
        gathered[r] = part
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ctx.fold_partials_dev(gathered.data_ptr(), world, out.data_ptr())
    got = bytes(out.cpu().numpy())
    if got != C.msm_pippenger(s, p, 1):
        bad += 1
        print("MISMATCH round %d n=%d big=%d: %s" % (it, n, nb, got.hex()))
print("tiny-after-big: %d rounds, %d mismatches" % (rounds, bad))
ctx.close()
