"""dev tool (GPU box): the 1 024-proof aggregation job with 8 jobs in flight on 8 contexts (bench.py's
aggregate_1024_proofs_pipelined), ms per job -- for A/B of library variants (SNARKV_AMD_LIB)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import snark_verifier_amd as sv

m, K = 1024, 8
streams = [torch.cuda.Stream() for _ in range(K)]
ctxs = [sv.Context(0, stream=s.cuda_stream) for s in streams]
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
dks = [sv.DecidingKey(c, g1, g2, g2) for c in ctxs]
offs = [0]
for _ in range(m):
    offs += [offs[-1] + 21, offs[-1] + 24]
n1, n2 = offs[-1], 2 * (m + 1)
ds = torch.empty(32 * max(n1, n2), dtype=torch.uint8, device="cuda")
dp = torch.empty(64 * max(n1, n2), dtype=torch.uint8, device="cuda")
ctxs[0].sample_scalars_dev(0x5EED0003, max(n1, n2), ds.data_ptr())
ctxs[0].sample_points_dev(0x5EED0004, max(n1, n2), dp.data_ptr())
o1 = torch.tensor(offs, dtype=torch.int32, device="cuda")
o2 = torch.tensor([0, m + 1, n2], dtype=torch.int32, device="cuda")
out1 = [torch.zeros(64 * (len(offs) - 1), dtype=torch.uint8, device="cuda") for _ in ctxs]
acc = [torch.zeros(128, dtype=torch.uint8, device="cuda") for _ in ctxs]
ok = [torch.zeros(1, dtype=torch.uint8, device="cuda") for _ in ctxs]
torch.cuda.synchronize()


def wave():
    for k, c in enumerate(ctxs):
        c.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o1.data_ptr(), len(offs) - 1, n1, out1[k].data_ptr())
        c.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o2.data_ptr(), 2, n2, acc[k].data_ptr())
        c.decide_batch_dev(dks[k], acc[k].data_ptr(), 1, ok[k].data_ptr())


for _ in range(2):
    wave()
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 6
for _ in range(R):
    wave()
torch.cuda.synchronize()
print("aggregate_1024_pipelined: %.4f ms per job" % ((time.perf_counter() - t0) / (R * K) * 1e3))
