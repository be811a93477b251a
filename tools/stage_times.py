"""Dev tool: per-stage HIP-event times of one 2^k-point Pippenger (sequential), averaged."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snark_verifier_amd as sv
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
ctx = sv.Context(0)
ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda"); dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
out = torch.zeros(64, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.sample_scalars_dev(1, n, ds.data_ptr()); ctx.sample_points_dev(2, n, dp.data_ptr()); ctx.sync()
ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), 0); ctx.sync()
ctx.set_stage_timing(True)
acc = {}
for _ in range(10):
    ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), 0)
    for kk, v in ctx.get_stage_timing().items():
        acc[kk] = acc.get(kk, 0.0) + v / 10
print({kk: round(v, 3) for kk, v in acc.items()})
