"""dev tool (GPU box): keep the bench's workload (batches of 20 MSMs of 2^20 points) running for --seconds, for
tools/clock_probe.sh to sample the clocks / power next to it.  Touches /tmp/load_loop.ready when the loop starts."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import snark_verifier_amd as sv

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=8.0)
a = ap.parse_args()
st = torch.cuda.Stream()
ctx = sv.Context(0, stream=st.cuda_stream)
n, k = 1 << 20, 20
ds = [torch.empty(32 * n, dtype=torch.uint8, device="cuda") for _ in range(k)]
dp = [torch.empty(64 * n, dtype=torch.uint8, device="cuda") for _ in range(k)]
for i in range(k):
    ctx.sample_scalars_dev(0x5EED0001, n, ds[i].data_ptr(), first=i * n)
    ctx.sample_points_dev(0x5EED0002, n, dp[i].data_ptr(), first=i * n)
out = torch.zeros(64 * k, dtype=torch.uint8, device="cuda")
ctx.sync()
args = ([t.data_ptr() for t in ds], [t.data_ptr() for t in dp], [n] * k, out.data_ptr())
ctx.msm_pippenger_many_dev(*args)
ctx.sync()
open("/tmp/load_loop.ready", "w").write("1")
t0, batches = time.perf_counter(), 0
while time.perf_counter() - t0 < a.seconds:
    ctx.msm_pippenger_many_dev(*args)
    ctx.sync()
    batches += 1
dt = time.perf_counter() - t0
print("# load loop: %d batches of %d MSMs in %.2f s = %.4f ms per MSM" % (batches, k, dt, dt / (batches * k) * 1e3))
