"""dev tool (GPU box): the aggregation job of bench.py's second metric (tools/aggregate_job.py) with N jobs in flight, one
context + stream each -- ms per job and proofs/s by N.  What the hardware-queue record was taken with:
  for q in 4 8 16 24; do GPU_MAX_HW_QUEUES=$q python tools/aggregate_inflight.py; done      (profiles/r03_agg_hw_queues.txt)
  rocprofv3 --kernel-trace --stats -- python tools/aggregate_inflight.py --proofs 1024 --inflight 16   (who stretches under overlap)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import snark_verifier_amd as sv

ap = argparse.ArgumentParser()
ap.add_argument("--proofs", type=int, nargs="*", default=[64, 1024])
ap.add_argument("--inflight", type=int, nargs="*", default=[1, 4, 8, 16, 32])
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--dummy-streams", type=int, default=0, help="streams created (and kept) before the jobs' own: does their queue mapping matter?")
ap.add_argument("--dummy-priority", type=int, default=0)
ap.add_argument("--hint-from", type=int, default=2, help="jobs in flight from which the contexts get the throughput hint")
a = ap.parse_args()
g2 = bytes.fromhex(
    "edf692d95cbdde46ddda5ef7d422436779445c5e66006a42761e1f12efde0018c212f3aeb785e49712e7a9353349aaf1255dfb31b7bf60723a480d9293938e19"
    "aa7dfa6601cce64c7bd3430c69e7d1e38f40cb8d8071ab4aeb6d8cdba55ec8125b9722d1dcdaac55f38eb37033314bbc95330c69ad999eec75f05f58d0890609")
g1 = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
dummies = [torch.cuda.Stream(priority=a.dummy_priority) for _ in range(a.dummy_streams)]
for d_ in dummies:
    with torch.cuda.stream(d_):
        torch.zeros(1, device="cuda")
nmax = max(a.inflight)
streams = [torch.cuda.Stream() for _ in range(nmax)]
ctxs = [sv.Context(0, stream=s.cuda_stream) for s in streams]
dks = [sv.DecidingKey(c, g1, g2, g2) for c in ctxs]
for nproofs in a.proofs:
    offs = [0]
    for _ in range(nproofs):
        offs += [offs[-1] + 21, offs[-1] + 24]
    n1, n2 = offs[-1], 2 * (nproofs + 1)
    ds = torch.empty(32 * max(n1, n2), dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * max(n1, n2), dtype=torch.uint8, device="cuda")
    ctxs[0].sample_scalars_dev(0x5EED0003, max(n1, n2), ds.data_ptr())
    ctxs[0].sample_points_dev(0x5EED0004, max(n1, n2), dp.data_ptr())
    o1 = torch.tensor(offs, dtype=torch.int32, device="cuda")
    o2 = torch.tensor([0, nproofs + 1, n2], dtype=torch.int32, device="cuda")
    out1 = [torch.zeros(64 * (len(offs) - 1), dtype=torch.uint8, device="cuda") for _ in ctxs]
    acc = [torch.zeros(128, dtype=torch.uint8, device="cuda") for _ in ctxs]
    ok = [torch.zeros(1, dtype=torch.uint8, device="cuda") for _ in ctxs]
    torch.cuda.synchronize()

    def job(k):
        c = ctxs[k]
        c.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o1.data_ptr(), len(offs) - 1, n1, out1[k].data_ptr())
        c.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o2.data_ptr(), 2, n2, acc[k].data_ptr())
        c.decide_batch_dev(dks[k], acc[k].data_ptr(), 1, ok[k].data_ptr())

    for N in a.inflight:
        for c in ctxs:  # as bench.py does: the library's throughput hint while several jobs share the GPU
            c.set_throughput_hint(N >= a.hint_from)

        def wave():
            for _ in range(a.rounds):
                for k in range(N):
                    job(k)

        wave()
        torch.cuda.synchronize()
        best, submit = 1e9, 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            wave()
            t1 = time.perf_counter()  # everything enqueued: the host's share (one Python thread, ctypes calls)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / (N * a.rounds) * 1e3)
            submit = min(submit, (t1 - t0) / (N * a.rounds) * 1e3)
        print("queues=%s proofs=%d inflight=%d throughput_hint=%d ms_per_job=%.4f proofs_per_s=%.3e host_submit_ms_per_job=%.4f" % (
            os.environ.get("GPU_MAX_HW_QUEUES"), nproofs, N, int(N >= a.hint_from), best, nproofs / best * 1e3, submit), flush=True)
for d in dks:
    d.close()
