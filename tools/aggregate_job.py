"""dev tool (GPU box): ONLY the aggregation job of bench.py's second metric (per proof a 21- and a 3-term MSM, then KzgAs'
two (m + 1)-term MSMs, then one pairing decide), `--reps` times, so that
  rocprofv3 --kernel-trace --stats -- python tools/aggregate_job.py --proofs 64
gives the kernel shares of that job alone (profiles/r03_rocprofv3_kernel_stats_aggregate_<proofs>.csv)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import snark_verifier_amd as sv

ap = argparse.ArgumentParser()
ap.add_argument("--proofs", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
st = torch.cuda.Stream()
ctx = sv.Context(0, stream=st.cuda_stream)
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
dk = sv.DecidingKey(ctx, g1, g2, g2)
m = a.proofs
offs = [0]
for _ in range(m):
    offs += [offs[-1] + 21, offs[-1] + 24]
n1, n2 = offs[-1], 2 * (m + 1)
ds = torch.empty(32 * max(n1, n2), dtype=torch.uint8, device="cuda")
dp = torch.empty(64 * max(n1, n2), dtype=torch.uint8, device="cuda")
ctx.sample_scalars_dev(0x5EED0003, max(n1, n2), ds.data_ptr())
ctx.sample_points_dev(0x5EED0004, max(n1, n2), dp.data_ptr())
o1 = torch.tensor(offs, dtype=torch.int32, device="cuda")
o2 = torch.tensor([0, m + 1, n2], dtype=torch.int32, device="cuda")
out1 = torch.zeros(64 * (len(offs) - 1), dtype=torch.uint8, device="cuda")
acc = torch.zeros(128, dtype=torch.uint8, device="cuda")
ok = torch.zeros(1, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def job():
    ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o1.data_ptr(), len(offs) - 1, n1, out1.data_ptr())
    ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o2.data_ptr(), 2, n2, acc.data_ptr())
    ctx.decide_batch_dev(dk, acc.data_ptr(), 1, ok.data_ptr())


for _ in range(3):
    job()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    job()
torch.cuda.synchronize()
print("aggregate_%d_proofs: %.4f ms per job" % (m, (time.perf_counter() - t0) / a.reps * 1e3))
dk.close()
