"""Dev tool: steady-state throughput of K MSMs in flight (no stage timing), 2^20 points each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snark_verifier_amd as sv
n = 1 << 20
base = sv.Context(0)
ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda"); dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
base.sample_scalars_dev(1, n, ds.data_ptr()); base.sample_points_dev(2, n, dp.data_ptr()); base.sync()
res = []
for K in (1, 2, 3, 4, 6, 8):
    ctxs = [sv.Context(0, ordered=False) for _ in range(K)]
    outs = [torch.zeros(64, dtype=torch.uint8, device="cuda") for _ in range(K)]
    reps = 24
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(reps):
            ctxs[i % K].msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, outs[i % K].data_ptr(), 0)
        for c in ctxs: c.sync()
        ms = (time.perf_counter() - t0) * 1e3 / reps
    res.append("K=%d %.3f" % (K, ms))
    assert all(bytes(o.cpu().numpy()) == bytes(outs[0].cpu().numpy()) for o in outs)
    for c in ctxs: c.close()
print("acc_stream=%s queues=%s ms/MSM:" % (os.environ.get("SNARKV_ACC_STREAM", "1"), os.environ.get("GPU_MAX_HW_QUEUES", "4")), "  ".join(res))
