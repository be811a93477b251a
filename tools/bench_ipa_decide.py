"""Dev tool: `IpaAs::decide` at k = 16..22 -- the device decider over a resident committing key
(snarkv_ipa_decide_batch: k scalars in, 64 bytes out) against the host route it replaces
(h_coeffs on the host, 96 B per term over PCIe into snarkv_g1_msm_pippenger).
Run on the GPU box: python tools/bench_ipa_decide.py"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import snark_verifier_amd as sv

ctx = sv.Context(0)
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
rnd = random.Random(1)
for k in (16, 18, 20, 22):
    n = 1 << k
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sample_points_dev(7, n, dp.data_ptr())
    ctx.sync()
    gb = bytes(dp.cpu().numpy())
    t0 = time.perf_counter()
    dk = sv.IpaDecidingKey(ctx, gb)
    t_up = (time.perf_counter() - t0) * 1e3
    xi = b"".join(rnd.randrange(R).to_bytes(32, "little") for _ in range(k))
    u = gb[:64]  # any point: the verdict is not what is timed
    ctx.ipa_decide_batch(dk, xi, u)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.ipa_decide_batch(dk, xi, u)
    t_dec = (time.perf_counter() - t0) * 1e3 / reps
    ctx.ipa_decide_batch(dk, xi * 8, u * 8)  # first call creates the lanes and their scratch
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.ipa_decide_batch(dk, xi * 8, u * 8)
    t_dec8 = (time.perf_counter() - t0) * 1e3 / 24
    # the host route: scalars + points cross PCIe every time (h_coeffs itself not even counted)
    hs = os.urandom(31 * n)
    hb = b"".join(hs[31 * i:31 * i + 31] + b"\x00" for i in range(0, n, max(1, n // 4096)))  # cheap filler
    hb = (hb * (n * 32 // len(hb) + 1))[:32 * n]
    ctx.msm_pippenger(hb, gb)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_pippenger(hb, gb)
    t_host = (time.perf_counter() - t0) * 1e3 / reps
    print({"k": k, "key_upload_ms": round(t_up, 2), "decide_ms": round(t_dec, 3), "decide_ms_in_batch_of_8": round(t_dec8, 3),
           "host_route_msm_only_ms": round(t_host, 3)})
    dk.close()
    del dp
