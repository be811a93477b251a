"""Dev tool: single-call latency of large MSMs, split (chunk-pipelined) vs unsplit (SNARKV_PIP_SPLIT=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snark_verifier_amd as sv
s = torch.cuda.Stream()
ctx = sv.Context(0, stream=s.cuda_stream)
for lg in (21, 22, 23, 24):
    n = 1 << lg
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda"); dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    ctx.sample_scalars_dev(1, n, ds.data_ptr()); ctx.sample_points_dev(2, n, dp.data_ptr())
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), 0)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(4):
        ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), 0)
    ctx.sync(); ms = (time.perf_counter() - t0) / 4 * 1e3
    print("split=%s 2^%d: %.2f ms  %.1f Mpts/s  %s" % (os.environ.get("SNARKV_PIP_SPLIT", "1"), lg, ms, n / ms / 1e3, bytes(out.cpu().numpy()).hex()[:12]))
    del ds, dp
