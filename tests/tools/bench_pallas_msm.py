"""Dev tool: throughput of the pasta build's MSM at 2^20 distinct pallas points (P_i = P_0 + i*Q,
made on the host: the library has no pallas sampler) -- one MSM at a time and 4 in flight, next to the
BN254 library on the same box.  Run on the GPU box: python tests/tools/bench_pallas_msm.py [log2n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch

import pallas as PA  # test infrastructure, used here only to MAKE input points
import snark_verifier_amd as sv
from snark_verifier_amd import pallas as PL

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
t0 = time.time()
P = PA.P
base, step = PA.sample_points(1, 2)
js, cur, q = [], PA._to_j(base), PA._to_j(step)
for _ in range(n):
    js.append(cur)
    cur = PA._jadd(cur, q)
# batched normalisation (one inversion)
pref, acc = [], 1
for x, y, z in js:
    pref.append(acc)
    acc = acc * z % P
inv = pow(acc, -1, P)
out = bytearray(64 * n)
for i in range(n - 1, -1, -1):
    x, y, z = js[i]
    zi = inv * pref[i] % P
    inv = inv * z % P
    zi2 = zi * zi % P
    out[64 * i:64 * i + 32] = (x * zi2 % P).to_bytes(32, "little")
    out[64 * i + 32:64 * i + 64] = (y * zi2 * zi % P).to_bytes(32, "little")
print("made %d pallas points in %.1f s" % (n, time.time() - t0))
scal = os.urandom(32 * n)
sb = bytearray(scal)
for i in range(n):
    sb[32 * i + 31] &= 0x3F  # < 2^254 < r
dp = torch.frombuffer(out, dtype=torch.uint8).cuda()
ds = torch.frombuffer(sb, dtype=torch.uint8).cuda()
res = {}
for name, mk in (("pallas", lambda s: PL.PallasContext(0, s)), ("bn254", lambda s: sv.Context(0, s))):
    if name == "bn254":  # same scalars; BN254 points from the library's sampler
        dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
        c0 = sv.Context(0)
        c0.sample_points_dev(2, n, dp.data_ptr())
        c0.sync()
    for K in (1, 4):
        streams = [torch.cuda.Stream() for _ in range(K)]
        ctxs = [mk(s.cuda_stream) for s in streams]
        outs = [torch.zeros(64, dtype=torch.uint8, device="cuda") for _ in range(K)]
        torch.cuda.synchronize()
        reps = 16
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                ctxs[i % K].msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, outs[i % K].data_ptr(), 0)
            for c in ctxs:
                c.sync()
            ms = (time.perf_counter() - t0) * 1e3 / reps
        assert all(bytes(o.cpu().numpy()) == bytes(outs[0].cpu().numpy()) for o in outs)
        res["%s_K%d_ms" % (name, K)] = round(ms, 3)
        for c in ctxs:
            c.close()
print(res)
