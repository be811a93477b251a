"""Worker of tests/test_gpu_two_ranks.py: ONE of two OS processes that share the box's GPU, each with its own HIP context.
Launched by torch.distributed.run (gloo: RCCL refuses one device twice).  Runs the product's multi-process wiring on real
kernels -- snark-verifier_amd/distributed.py over the C ABI -- and prints one JSON line of results per rank."""
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import snark_verifier_amd as sv
from snark_verifier_amd import distributed as D
from snark_verifier_amd import host_api as HA


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)  # every rank on the SAME device: two contexts, two sets of streams, one GPU
    st = torch.cuda.Stream()
    ctx = sv.Context(0, stream=st.cuda_stream)
    out = {"rank": rank, "world": world, "pid": os.getpid()}
    # ---- K point-sharded MSMs, ONE exchange of K x 144 B per rank, K folds in one launch on every rank
    totals = [200_000, 4097, 1]  # the last one leaves rank 1's shard EMPTY (ceil chunking)
    ds, dp, counts = [], [], []
    with torch.cuda.stream(st):
        for i, n in enumerate(totals):
            lo, hi = D.shard_range(n, rank, world)
            m = max(hi - lo, 1)
            s = torch.empty(32 * m, dtype=torch.uint8, device="cuda")
            p = torch.empty(64 * m, dtype=torch.uint8, device="cuda")
            ctx.sample_scalars_dev(0x7A00 + i, m, s.data_ptr(), first=lo)
            ctx.sample_points_dev(0x7B00 + i, m, p.data_ptr(), first=lo)
            ds.append(s), dp.append(p), counts.append(hi - lo)
        res = D.gpu_sharded_msm_batch(ctx, ds, dp, counts, stream=st)
        ctx.sync()
        out["batch"] = bytes(res.cpu().numpy()).hex()
        # the single-MSM form and the bucket-sharded form on MSM 0
        out["single"] = bytes(D.gpu_sharded_msm(ctx, ds[0], dp[0], totals[0]).cpu().numpy()).hex()
        ctx.sync()
        out["bucket_sharded"] = bytes(D.gpu_bucket_sharded_msm(ctx, ds[0], dp[0], totals[0]).cpu().numpy()).hex()
        ctx.sync()
    # ---- proof-sharded aggregation of the committed 64-proof fixture; then with one proof of RANK 1's shard corrupted
    fx = HA.read_fixture(os.path.join(ROOT, "tests", "golden", "bench_plonk_gwc19_evm_64.bin"))
    n, ib, prb = fx["n"], fx["instances"], fx["proofs"]
    insts, proofs, off = [], [], 0
    for _ in range(n):
        cols, = struct.unpack_from("<I", ib, off)
        o2 = off + 4
        for _ in range(cols):
            m, = struct.unpack_from("<I", ib, o2)
            o2 += 4 + 32 * m
        insts.append(ib[off:o2])
        off = o2
    off = 0
    for _ in range(n):
        ln, = struct.unpack_from("<I", prb, off)
        proofs.append(prb[off + 4:off + 4 + ln])
        off += 4 + ln
    hp, hdk = HA.Protocol(fx["protocol"]), HA.DecidingKey(fx["dk"])
    acc, ok = D.gpu_sharded_aggregation(hp, hdk, insts, proofs)
    out["agg_ok"], out["agg_acc_matches_fixture"] = bool(ok), acc == fx["expected_acc"]
    bad = list(proofs)
    bad[n - 2] = bad[n - 2][:100]  # in the LAST rank's shard: the reject must reach every rank
    out["agg_bad"] = list(D.gpu_sharded_aggregation(hp, hdk, insts, bad)) == [None, False]
    hp.close(), hdk.close()
    ctx.close()
    dist.barrier()
    print("RANKLINE " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
