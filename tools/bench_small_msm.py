"""Times the segmented naive MSM (snarkv_g1_msm_batched_dev) on the shapes of the
aggregation jobs, for the chunk count given in SNARKV_NAIVE_CHUNKS (dev tool)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import snark_verifier_amd as sv

ctx = sv.Context(0, stream=torch.cuda.current_stream().cuda_stream)
shapes = [("64 proofs: 128 segs/1536 terms", 64, 24), ("KzgAs 2x65", 2, 65), ("1024 proofs: 2048 segs/24576 terms", 1024, 24),
          ("KzgAs 2x1025", 2, 1025), ("1 MSM x 21", 1, 21), ("4096 terms/2", 2, 2048), ("8192 terms/2", 2, 4096)]
res = {}
for name, nseg, per in shapes:
    if per == 24:
        offs = [0]
        for _ in range(nseg):
            offs += [offs[-1] + 21, offs[-1] + 24]
    else:
        offs = [i * per for i in range(nseg + 1)]
    n = offs[-1]
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    ctx.sample_scalars_dev(1, n, ds.data_ptr())
    ctx.sample_points_dev(2, n, dp.data_ptr())
    o = torch.tensor(offs, dtype=torch.int32, device="cuda")
    out = torch.zeros(64 * (len(offs) - 1), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o.data_ptr(), len(offs) - 1, n, out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o.data_ptr(), len(offs) - 1, n, out.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    import hashlib
    res[name] = (round(ms, 3), hashlib.sha256(bytes(out.cpu().numpy())).hexdigest()[:12])
print("chunks=%s" % os.environ.get("SNARKV_NAIVE_CHUNKS", "auto"), res)
