"""Single-GPU emulation of the per-rank work of the two multi-GPU MSM shardings
 at world = 8: what one rank computes, plus the bytes it would exchange.
  point-sharded : full Pippenger of n/8... (weak scaling: n per rank) -> 144 B all-gather
  bucket-sharded: fill (n per rank, c of the total) + 7 bucket adds of its window share
                  + reduce of W/8 windows; exchange = the whole grid once (all-to-all)
Run on the GPU box: python tools/bench_bucket_sharded.py [log2_n_per_rank] [world]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import snark_verifier_amd as sv
from snark_verifier_amd.distributed import shard_range

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << k
st = torch.cuda.Stream()
ctx = sv.Context(0, st.cuda_stream)
ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.sample_scalars_dev(1, n, ds.data_ptr())
ctx.sample_points_dev(2, n, dp.data_ptr())
PB = sv.G1_PARTIAL_BYTES
c, W, B = sv.Context.bucket_geometry(n * world)
grid = torch.zeros(W * B * PB, dtype=torch.uint8, device="cuda")
other = torch.zeros(W * B * PB, dtype=torch.uint8, device="cuda")
part = torch.zeros(PB, dtype=torch.uint8, device="cuda")
w0, w1 = shard_range(W, 0, world)
own = (w1 - w0) * B


def timed(fn, reps=10):
    fn()
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t) / reps * 1e3


t_point = timed(lambda: ctx.msm_pippenger_partial_dev(ds.data_ptr(), dp.data_ptr(), n, part.data_ptr(), 0))
t_fill = timed(lambda: ctx.fill_buckets_dev(ds.data_ptr(), dp.data_ptr(), n, c, grid.data_ptr()))
ctx.fill_buckets_dev(ds.data_ptr(), dp.data_ptr(), n, c, other.data_ptr())
t_add = timed(lambda: [ctx.buckets_add_dev(grid.data_ptr(), other.data_ptr(), own) for _ in range(world - 1)])
t_red = timed(lambda: ctx.buckets_reduce_dev(grid.data_ptr(), c, w0, w1 - w0, part.data_ptr()))
xbytes = W * B * PB * (world - 1) / world
print({"n_per_rank": n, "world": world, "c_total": c, "windows": W, "grid_MB": W * B * PB / 1e6,
       "point_sharded_ms": round(t_point, 3), "bucket_fill_ms": round(t_fill, 3), "bucket_adds_ms": round(t_add, 3),
       "bucket_reduce_ms": round(t_red, 3), "bucket_total_compute_ms": round(t_fill + t_add + t_red, 3),
       "exchange_MB_per_rank": round(xbytes / 1e6, 1),
       "exchange_ms_at_7x153GBps": round(xbytes / (7 * 153e9) * 1e3, 3)})
