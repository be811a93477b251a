"""Multi-GPU MSM: point-sharded Pippenger + all-gather of projective partials.

MSM is linear, so the sum over points splits into per-rank sub-sums -- exactly
the reference's rayon chunking (`chunk = ceil(n / threads)`, each chunk runs the
whole serial Pippenger, results added: snark-verifier/src/util/msm.rs:311-336),
with GPUs in place of threads.  Each rank reduces ITS contiguous shard to one
projective partial (144 B), the partials are all-gathered (RCCL over xGMI when
the backend is "nccl"; 144 B per rank -> latency-bound, link bandwidth is
irrelevant) and every rank folds them locally, so all ranks hold the identical
affine result (all-reduce semantics; `ncclAllReduce` itself cannot be used: EC
addition is not an RCCL reduction operator).  SURVEY.md section 8(e).

One process per GPU; `torch.distributed` is only the transport.

Stream ordering.  The context enqueues on ITS stream (a private non-blocking one unless it was created on the caller's);
torch allocates, fills and runs the collective on torch's CURRENT stream.  Every step below that hands a buffer from one
to the other is bracketed by `ctx.wait_stream()` (the context's next launch runs after what torch has queued: fills,
the landed collective) and `ctx.stream_wait()` (torch's next op -- the collective, a copy, the caching allocator's reuse
of a freed block -- runs after what the context has queued): two event record/wait pairs, no host synchronisation
(include/snarkv_amd.h `snarkv_ctx_wait_stream` / `snarkv_stream_wait_ctx`).  A context created on torch's current stream
makes both a no-op.  The results these helpers return are therefore ordered on torch's current stream: `.cpu()` or
any torch op on them is safe; `ctx.sync()` alone is NOT needed (and would not be enough for torch-side consumers).
"""
from dataclasses import dataclass
from typing import Callable


_DATA_GROUP = None  # process group the device payloads travel on (None: the default group)


def set_data_group(group):
    """Route the device-side all-gathers over `group` (e.g. an RCCL group next to a gloo default group that carries the
    control traffic: barriers, timings, the fallback agreement of bench.py).  None restores the default group."""
    global _DATA_GROUP
    _DATA_GROUP = group


def all_gather_bytes(part):
    """every rank's `part` (uint8, same length), concatenated in rank order, on `part`'s device.  RCCL ("nccl") gathers device
    tensors directly; any other backend (gloo: CPU ranks, or several processes sharing ONE GPU, which RCCL refuses) is
    host-staged -- the payloads here are 144-byte partials and 128-byte accumulators."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    group = _DATA_GROUP if part.device.type == "cuda" else None
    if part.device.type != "cuda" or dist.get_backend(group) == "nccl":
        out = torch.empty(world * part.numel(), dtype=torch.uint8, device=part.device)
        dist.all_gather_into_tensor(out, part, group=group)
        return out
    torch.cuda.current_stream().synchronize()
    host = part.cpu()
    out = torch.empty(world * host.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, host, group=group)
    return out.to(part.device)


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank `rank`: chunk = ceil(n / world), as
    `Integer::div_ceil(&scalars.len(), &num_threads)` (msm.rs:322)."""
    chunk = -(-n_total // world)
    lo = min(rank * chunk, n_total)
    hi = min(lo + chunk, n_total)
    return lo, hi


@dataclass
class ShardedMsm:
    """`partial_fn(lo, hi) -> tensor[partial_bytes]` reduces the local shard;
    `fold_fn(tensor[world * partial_bytes], world) -> tensor[64]` folds the
    gathered partials.  On a GPU box these are the HIP entry points
    (`Context.msm_pippenger_partial_dev`, `Context.fold_partials_dev`); the gloo
    CPU tests inject oracle-backed doubles to exercise the plumbing."""

    partial_fn: Callable
    fold_fn: Callable
    partial_bytes: int = 144

    def run(self, n_total: int):
        import torch
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        lo, hi = shard_range(n_total, rank, world)
        part = self.partial_fn(lo, hi)
        assert part.numel() == self.partial_bytes and part.dtype == torch.uint8
        gathered = part if world == 1 else all_gather_bytes(part)
        return self.fold_fn(gathered, world)


def gpu_msm_partial(ctx, part, d_scalars, d_points, count, window_bits=0):
    """This rank's 144-byte projective partial of its shard (`count` points at the head of d_scalars / d_points) into
    `part`.  An EMPTY shard contributes the identity: ceil chunking leaves trailing ranks empty when n is small against
    the world size (n = 9, world = 8: ranks 5-7); the device call would return SNARKV_ERR_EMPTY on that rank only and
    the peers would hang in the all-gather."""
    if count == 0:
        part.zero_()  # all-zero partial: ZZ = 0 is the identity (on torch's stream, where the collective runs)
        return part
    ctx.wait_stream()  # `part`'s allocation / fill and the inputs (torch's stream) before the context's writes
    ctx.msm_pippenger_partial_dev(d_scalars.data_ptr(), d_points.data_ptr(), count, part.data_ptr(), window_bits)
    ctx.stream_wait()  # the partial is written before the all-gather (torch's stream) reads it
    return part


def gpu_sharded_msm(ctx, d_scalars, d_points, n_total, window_bits=0):
    """Product wiring: shard -> HIP Pippenger partial -> all_gather -> HIP fold.
    `d_scalars` / `d_points` hold THIS rank's shard only (weak scaling)."""
    import torch

    from . import G1_PARTIAL_BYTES

    part = torch.zeros(G1_PARTIAL_BYTES, dtype=torch.uint8, device=d_scalars.device)
    out = torch.zeros(64, dtype=torch.uint8, device=d_scalars.device)

    # The context's HIP stream need not be torch's current stream (a context
    # created without a stream owns a private one), and the collective is ordered
    # against torch's current stream only -- so order the three steps explicitly (module docstring).
    def partial_fn(lo, hi):
        return gpu_msm_partial(ctx, part, d_scalars, d_points, hi - lo, window_bits)

    def fold_fn(gathered, world):
        ctx.wait_stream()  # the all-gather has landed (and `out` is zero-filled) before the fold reads / writes
        ctx.fold_partials_dev(gathered.data_ptr(), world, out.data_ptr())
        ctx.stream_wait()  # `out` is valid for whatever torch's stream does with it; `gathered` may be freed
        return out

    return ShardedMsm(partial_fn, fold_fn, G1_PARTIAL_BYTES).run(n_total)


@dataclass
class ShardedMsmBatch:
    """K independent MSMs, every one sharded by points over the ranks, with ONE collective for the whole batch:
    `partials_fn(ranges) -> tensor[K * partial_bytes]` reduces this rank's shard of every MSM (`ranges[i]` = its
    (lo, hi) of MSM i), the ranks all-gather K x partial_bytes each, and `fold_fn(tensor[K][world][partial_bytes], world,
    K) -> tensor[K * 64]` folds per MSM.  On a GPU box these are `Context.msm_pippenger_many_partial_dev` and
    `Context.fold_partials_many_dev` (`gpu_sharded_msm_batch`, what `bench.py --gpus N` times); the gloo CPU tests inject
    oracle-backed doubles."""

    partials_fn: Callable
    fold_fn: Callable
    partial_bytes: int = 144

    def run(self, n_totals):
        """`n_totals[i]` = points of MSM i over ALL ranks: this rank reduces `shard_range(n_totals[i], rank, world)`."""
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        return self._gather_fold(self.partials_fn([shard_range(n, rank, world) for n in n_totals]), len(n_totals))

    def run_local(self, counts):
        """Weak-scaling form: the caller already holds THIS rank's shard of every MSM (`counts[i]` points of it, 0
        allowed: an empty shard contributes the identity partial); `partials_fn` receives `(0, counts[i])`."""
        return self._gather_fold(self.partials_fn([(0, int(c)) for c in counts]), len(counts))

    def _gather_fold(self, parts, k):
        import torch
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_initialized() else 1
        assert parts.numel() == k * self.partial_bytes and parts.dtype == torch.uint8
        gathered = parts if world == 1 else all_gather_bytes(parts)  # [rank][job][partial]
        by_job = gathered.view(world, k, self.partial_bytes).transpose(0, 1).contiguous()  # [job][rank][partial]
        return self.fold_fn(by_job, world, k)


def gpu_sharded_msm_batch(ctx, d_scalars, d_points, counts, window_bits=0, stream=None):
    """Product wiring of ShardedMsmBatch: `d_scalars[i]` / `d_points[i]` hold THIS rank's shard of MSM i (`counts[i]`
    points; 0 = an empty shard, which contributes the identity).  The launches go to the context's stream, the collective
    to `stream` (default: torch's current stream); the two are ordered by events (module docstring) -- free when the
    context was created on that very stream (`stream=torch_stream.cuda_stream`), which is what bench.py does.  Returns
    the device tensor of the K affine results (64 B each), valid on `stream`."""
    import contextlib

    import torch

    from . import G1_PARTIAL_BYTES

    k = len(counts)
    keep = []

    def partials_fn(ranges):
        # An empty local shard (n < world with the reference's ceil chunking, or a caller that passes 0) must not reach the
        # device call: it returns SNARKV_ERR_EMPTY on THAT rank only and every peer would block in the all-gather.  Its
        # 144-byte slot stays zero (ZZ = 0 is the identity partial); the other jobs go to the device as one batch.
        parts = torch.zeros(G1_PARTIAL_BYTES * k, dtype=torch.uint8, device=d_scalars[0].device)
        keep.append(parts)
        live = [i for i, (lo, hi) in enumerate(ranges) if hi > lo]
        ctx.wait_stream()  # the zero fill of `parts` and the caller's inputs before the context's launches
        if len(live) == k:
            ctx.msm_pippenger_many_partial_dev([t.data_ptr() for t in d_scalars], [t.data_ptr() for t in d_points],
                                               [hi - lo for lo, hi in ranges], parts.data_ptr(), window_bits)
        elif live:
            dense = torch.zeros(G1_PARTIAL_BYTES * len(live), dtype=torch.uint8, device=parts.device)
            keep.append(dense)
            ctx.wait_stream()
            ctx.msm_pippenger_many_partial_dev([d_scalars[i].data_ptr() for i in live], [d_points[i].data_ptr() for i in live],
                                               [ranges[i][1] - ranges[i][0] for i in live], dense.data_ptr(), window_bits)
            ctx.stream_wait()
            idx = torch.tensor(live, dtype=torch.int64, device=parts.device)
            parts.view(k, G1_PARTIAL_BYTES).index_copy_(0, idx, dense.view(len(live), G1_PARTIAL_BYTES))
        ctx.stream_wait()  # the partials are written before the all-gather reads them
        return parts

    def fold_fn(by_job, world, k_):
        out = torch.zeros(64 * k_, dtype=torch.uint8, device=by_job.device)
        keep.append(by_job)
        ctx.wait_stream()  # the gathered partials (and their transpose) have landed
        ctx.fold_partials_many_dev(by_job.data_ptr(), world, k_, out.data_ptr())
        ctx.stream_wait()
        return out

    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        res = ShardedMsmBatch(partials_fn, fold_fn, G1_PARTIAL_BYTES).run_local(list(counts))
    res._keep_alive = keep  # the partials and the transposed gather buffer outlive the asynchronous launches that read them
                            # (all allocated while `stream` is current: the caching allocator ties them to it)
    return res


@dataclass
class BucketShardedMsm:
    """The "bucket-sum allreduce" alternative of SURVEY.md 8(e): ranks still hold disjoint
    POINT shards, but instead of finishing their own Pippenger they all fill the same
    global bucket grid (`windows x buckets_per_window`, the `buckets[d-1].add_assign`
    half of msm.rs:291-296), exchange it by WINDOW range (an all-to-all: rank g receives
    every rank's copy of the windows g owns -- a reduce-scatter whose reduction is EC
    addition, which RCCL cannot do in-flight), add the copies, and run the running-sum +
    `double()` half (msm.rs:285-302) on their windows only.  The per-rank projective
    partials are then all-gathered and folded as in `ShardedMsm`.

    Against point-sharding it saves (1 - 1/world) of the bucket-reduce work per rank and
    costs an exchange of the whole grid (windows * buckets * 144 B, tens of MB, against
    144 B per rank): see DESIGN.md section 6 for the measured trade.

    `fill_fn(lo, hi) -> uint8[windows * B * bucket_bytes]` (window-major),
    `add_fn(dst, src)` in place, `reduce_fn(buckets, w0, wcount) -> uint8[partial_bytes]`,
    `fold_fn(gathered, world) -> uint8[64]`."""

    windows: int
    buckets_per_window: int
    fill_fn: Callable
    add_fn: Callable
    reduce_fn: Callable
    fold_fn: Callable
    bucket_bytes: int = 144
    partial_bytes: int = 144

    def run(self, n_total: int):
        import torch
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        lo, hi = shard_range(n_total, rank, world)
        wbytes = self.buckets_per_window * self.bucket_bytes
        grid = self.fill_fn(lo, hi)
        assert grid.numel() == self.windows * wbytes and grid.dtype == torch.uint8
        w0, w1 = shard_range(self.windows, rank, world)
        if world == 1:
            mine = grid
        else:
            counts = [shard_range(self.windows, r, world) for r in range(world)]
            own = (w1 - w0) * wbytes
            splits = [(b - a) * wbytes for a, b in counts]
            if grid.device.type != "cuda" or dist.get_backend() == "nccl":
                recv = torch.empty(world * own, dtype=torch.uint8, device=grid.device)
                dist.all_to_all_single(recv, grid, output_split_sizes=[own] * world, input_split_sizes=splits)
            else:  # host-staged (gloo with device-resident grids: several processes on one GPU)
                torch.cuda.current_stream().synchronize()
                recv_h = torch.empty(world * own, dtype=torch.uint8)
                dist.all_to_all_single(recv_h, grid.cpu(), output_split_sizes=[own] * world, input_split_sizes=splits)
                recv = recv_h.to(grid.device)
            mine = recv[:own]
            for r in range(1, world):
                if own:
                    self.add_fn(mine, recv[r * own:(r + 1) * own])
        if w1 > w0:
            part = self.reduce_fn(mine, w0, w1 - w0)
        else:  # more ranks than windows: this rank contributes the identity
            part = torch.zeros(self.partial_bytes, dtype=torch.uint8, device=grid.device)
        assert part.numel() == self.partial_bytes
        gathered = part if world == 1 else all_gather_bytes(part)
        return self.fold_fn(gathered, world)


def gpu_bucket_sharded_msm(ctx, d_scalars, d_points, n_total, window_bits=0):
    """Product wiring of `BucketShardedMsm` over the HIP entry points.  `d_scalars` /
    `d_points` hold THIS rank's shard only; the window size is the one of the TOTAL n so
    that all ranks fill the same grid."""
    import torch

    from . import G1_PARTIAL_BYTES, Context

    c, windows, bpw = Context.bucket_geometry(n_total, window_bits)
    dev = d_scalars.device
    grid = torch.empty(windows * bpw * G1_PARTIAL_BYTES, dtype=torch.uint8, device=dev)
    part = torch.zeros(G1_PARTIAL_BYTES, dtype=torch.uint8, device=dev)
    out = torch.zeros(64, dtype=torch.uint8, device=dev)

    def fill_fn(lo, hi):
        if hi > lo:
            ctx.wait_stream()  # the inputs and the grid's allocation before the fill
            ctx.fill_buckets_dev(d_scalars.data_ptr(), d_points.data_ptr(), hi - lo, c, grid.data_ptr())
            ctx.stream_wait()  # the grid is written before the all-to-all reads it
        else:
            grid.zero_()
        return grid

    def add_fn(dst, src):
        ctx.wait_stream()  # the exchange has landed
        ctx.buckets_add_dev(dst.data_ptr(), src.data_ptr(), dst.numel() // G1_PARTIAL_BYTES)
        ctx.stream_wait()

    def reduce_fn(buckets, w0, wcount):
        ctx.wait_stream()
        ctx.buckets_reduce_dev(buckets.data_ptr(), c, w0, wcount, part.data_ptr())
        ctx.stream_wait()
        return part

    def fold_fn(gathered, world):
        ctx.wait_stream()
        ctx.fold_partials_dev(gathered.data_ptr(), world, out.data_ptr())
        ctx.stream_wait()
        return out

    return BucketShardedMsm(windows, bpw, fill_fn, add_fn, reduce_fn, fold_fn, G1_PARTIAL_BYTES,
                            G1_PARTIAL_BYTES).run(n_total)


@dataclass
class ShardedIpaDecide:
    """`IpaAs::decide` (reference pcs/ipa/decider.rs:47-55) with the 2^k-point committing key sharded over
    the ranks: the check U == <h(xi), G> is one MSM, linear in the points, so rank g commits to its slice
    G[lo:hi] with the matching slice of h_coeffs (`partial_fn(lo, hi, xi) -> uint8[partial_bytes]`), the
    projective partials are all-gathered (144 B per rank, latency-bound) and every rank folds them
    (`fold_fn(gathered, world) -> uint8[64]`) and compares with U -- all ranks hold the same verdict."""

    k: int
    partial_fn: Callable
    fold_fn: Callable
    partial_bytes: int = 144

    def run(self, xi, u) -> bool:
        import torch
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        lo, hi = shard_range(1 << self.k, rank, world)
        if hi > lo:
            part = self.partial_fn(lo, hi, xi)
        else:  # more ranks than points: the identity
            part = torch.zeros(self.partial_bytes, dtype=torch.uint8, device="cuda" if world > 1 and dist.get_backend() == "nccl" else "cpu")
        assert part.numel() == self.partial_bytes and part.dtype == torch.uint8
        gathered = part if world == 1 else all_gather_bytes(part)
        got = self.fold_fn(gathered, world)
        return bytes(got.cpu().numpy()) == bytes(u)


def gpu_sharded_ipa_decide(ctx, dk_shard, xi, u):
    """Product wiring: `dk_shard` = IpaDecidingKey(ctx, g[lo:hi], k, lo) of THIS rank (shard_range(2^k, rank, world));
    xi = k scalars (32 bytes each), u = 64 bytes.  -> bool, identical on every rank."""
    import torch

    from . import G1_PARTIAL_BYTES

    part = torch.zeros(G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")

    def partial_fn(lo, hi, xi_):
        ctx.wait_stream()
        ctx.ipa_commit_partial_dev(dk_shard, xi_, part.data_ptr())
        ctx.stream_wait()  # the partial is written before the all-gather reads it
        return part

    def fold_fn(gathered, world):
        ctx.wait_stream()
        ctx.fold_partials_dev(gathered.data_ptr(), world, out.data_ptr())
        ctx.stream_wait()
        return out

    return ShardedIpaDecide(dk_shard.k, partial_fn, fold_fn, G1_PARTIAL_BYTES).run(xi, u)


@dataclass
class ShardedAggregation:
    """Proof-sharded aggregation: proofs are independent,
    so rank g succinct-verifies ITS contiguous shard of the proofs
    (`verify_fn(lo, hi) -> bytes`, 128 bytes per accumulator), the accumulators are
    all-gathered (a few KiB: latency-bound) and every rank folds them and decides
    (`combine_fn(all_accumulator_bytes) -> (acc128, ok)`), so all ranks hold the same
    verdict.  No other exchange: the per-proof MSMs never leave their GPU.

    A shard that fails to verify (an invalid proof is a NORMAL outcome: `verify_fn` raises or returns None)
    must not leave its rank outside the collectives while the peers block in them: the failure travels as
    length -1 in the size exchange and EVERY rank returns `(None, False)` after it."""

    verify_fn: Callable
    combine_fn: Callable

    def run(self, n_proofs: int):
        import torch
        import torch.distributed as dist

        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        lo, hi = shard_range(n_proofs, rank, world)
        failed, self.last_error = False, None
        local = b""
        if hi > lo:
            try:
                got = self.verify_fn(lo, hi)
                if got is None:
                    failed = True
                else:
                    local = bytes(got)
            except Exception as e:  # reject on this shard: reported to every rank below
                failed, self.last_error = True, e
        assert len(local) % 128 == 0
        if world == 1:
            return (None, False) if failed else self.combine_fn(local)
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        mine = torch.tensor([-1 if failed else len(local)], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, mine)
        if int(sizes.min().item()) < 0:  # some shard rejected: same answer on every rank, no further collective
            return None, False
        cap = int(sizes.max().item())
        buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
        if local:
            buf[: len(local)] = torch.frombuffer(bytearray(local), dtype=torch.uint8).to(dev)
        gathered = torch.zeros(world * cap, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, buf)
        g = bytes(gathered.cpu().numpy())
        allb = b"".join(g[r * cap: r * cap + int(sizes[r].item())] for r in range(world))  # rank order = proof order
        return self.combine_fn(allb)


def gpu_sharded_aggregation(protocol, dk, instances_packed, proofs, mos=0, transcript=0):
    """Product wiring over the host mirror's C API (libsnarkv_host.so, include/snarkv_host.h; `host_api.Protocol` /
    `host_api.DecidingKey` handles): `instances_packed[i]` and `proofs[i]` are the i-th proof's packed instances /
    proof bytes; returns (acc128, ok) -- `(None, False)` on every rank when any shard fails to verify."""
    from . import host_api as H

    def verify_fn(lo, hi):
        ib = b"".join(instances_packed[lo:hi])
        return H.plonk_succinct_verify_batch(protocol, dk, ib, H.pack_proofs(proofs[lo:hi]), hi - lo, mos, transcript)

    def combine_fn(allb):
        # the accumulation transcript is of the proofs' family, as host/aggregation.hpp (Keccak | Poseidon)
        acc = H.kzg_as_create_proof(allb, H.TRANSCRIPT_EVM if transcript == H.TRANSCRIPT_EVM else H.TRANSCRIPT_POSEIDON)[0]
        return acc, H.kzg_decide(dk, acc)

    return ShardedAggregation(verify_fn, combine_fn).run(len(proofs))
