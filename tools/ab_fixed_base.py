#!/usr/bin/env python3
"""A/B of the fixed-base rows (csrc/msm_fixed.hip) against the plain segmented MSM on the shapes of the aggregation job's
first stage: per proof a 21-term and a 3-term MSM; with the table 9 of the 21 bases are fixed.  Prints ms per launch set
(mean of back-to-back calls) for 64 / 1 024 / 16 x 1 024 proofs.  Under rocprofv3 --kernel-trace --stats the per-kernel
durations tell where the time goes.   python tools/ab_fixed_base.py [--proofs 64 1024 16384] [--reps 10]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

import snark_verifier_amd as sv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--proofs", type=int, nargs="+", default=[64, 1024, 16384])
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--hint", action="store_true", help="throughput hint on the context")
ap.add_argument("--only", default="abcd", help="which variants to run (a plain, b mixed, c the variable terms alone, d the fixed terms alone)")
args = ap.parse_args()
ctx = sv.Context(0)
if args.hint:
    ctx.set_throughput_hint(True)
fb = torch.empty(64 * 9, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.sample_points_dev(0x5EED0006, 9, fb.data_ptr())
ctx.sync()
tab = sv.FixedTable(ctx, bytes(fb.cpu().numpy()))


def t_ms(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for m in args.proofs:
    offs, voffs, foffs = [0], [0], [0]
    for _ in range(m):
        offs += [offs[-1] + 21, offs[-1] + 24]
        voffs += [voffs[-1] + 12, voffs[-1] + 15]
        foffs += [foffs[-1] + 9, foffs[-1] + 9]
    n1, nv, nf = offs[-1], voffs[-1], foffs[-1]
    ds = torch.empty(32 * n1, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n1, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sample_scalars_dev(0x5EED0003, n1, ds.data_ptr())
    ctx.sample_points_dev(0x5EED0004, n1, dp.data_ptr())
    ctx.sync()
    o1 = torch.tensor(offs, dtype=torch.int32, device="cuda")
    vo = torch.tensor(voffs, dtype=torch.int32, device="cuda")
    fo = torch.tensor(foffs, dtype=torch.int32, device="cuda")
    fid = (torch.arange(nf, dtype=torch.int32, device="cuda") % 9).contiguous()
    out = torch.zeros(64 * 2 * m, dtype=torch.uint8, device="cuda")
    only_fixed_offs = torch.zeros(2 * m + 1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    a = b = c = d = float("nan")
    if "a" in args.only:
        a = t_ms(lambda: ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), o1.data_ptr(), 2 * m, n1, out.data_ptr()), args.reps)
    if "b" in args.only:
      b = t_ms(lambda: ctx.msm_batched_fixed_dev(tab, ds.data_ptr(), dp.data_ptr(), vo.data_ptr(), nv, ds.data_ptr() + 32 * nv,
                                               fid.data_ptr(), fo.data_ptr(), nf, 2 * m, out.data_ptr()), args.reps)
    if "c" in args.only:
        c = t_ms(lambda: ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), vo.data_ptr(), 2 * m, nv, out.data_ptr()), args.reps)
    if "d" in args.only:
      d = t_ms(lambda: ctx.msm_batched_fixed_dev(tab, 0, 0, only_fixed_offs.data_ptr(), 0, ds.data_ptr() + 32 * nv, fid.data_ptr(),
                                               fo.data_ptr(), nf, 2 * m, out.data_ptr()), args.reps)
    print("proofs %6d: plain 24 terms/proof %.3f ms | 15 variable + 9 fixed %.3f ms | the 15 variable alone %.3f ms | the 9 fixed alone %.3f ms"
          % (m, a, b, c, d), flush=True)
