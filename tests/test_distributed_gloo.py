"""CPU, world_size 2, gloo: the multi-GPU plumbing of the sharded MSM
(snark-verifier_amd/distributed.py) -- shard ranges, all-gather order, every
rank folding to the same result -- with oracle-backed doubles standing in for
the two HIP entry points (there is no GPU here)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    import coracle as C
    from snark_verifier_amd.distributed import ShardedMsm, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, p = C.sample_scalars(7, n), C.sample_points(8, n)

    def partial_fn(lo, hi):
        # oracle double: the shard's MSM as affine bytes, zero-padded to 144
        if hi > lo:
            pt = C.msm_pippenger(s[32 * lo:32 * hi], p[64 * lo:64 * hi], 1)
        else:
            pt = b"\x00" * 64
        return torch.frombuffer(bytearray(pt + b"\x00" * 80), dtype=torch.uint8)

    def fold_fn(gathered, w):
        acc = b"\x00" * 64
        raw = bytes(gathered.numpy())
        for k in range(w):
            acc = C.g1_add(acc, raw[144 * k:144 * k + 64])
        return torch.frombuffer(bytearray(acc), dtype=torch.uint8)

    out = ShardedMsm(partial_fn, fold_fn).run(n)
    q.put((rank, bytes(out.numpy()), shard_range(n, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 2, 101, 1000])
def test_sharded_msm_two_ranks_agree_with_single_process(n):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as C

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    expected = C.msm_pippenger(C.sample_scalars(7, n), C.sample_points(8, n), 1)
    ranges = sorted(r[2] for r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == n and ranges[0][1] == ranges[1][0]  # disjoint cover
    for _, out, _ in res:
        assert out == expected  # all ranks hold the same (all-reduce semantics)


def _batch_worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    import coracle as C
    from snark_verifier_amd.distributed import ShardedMsmBatch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    jobs = [(C.sample_scalars(70 + i, n), C.sample_points(80 + i, n)) for i, n in enumerate(sizes)]

    def partials_fn(ranges):  # oracle double: every shard's MSM as affine bytes, zero-padded to 144
        buf = bytearray()
        for (s, p), (lo, hi) in zip(jobs, ranges):
            pt = C.msm_pippenger(s[32 * lo:32 * hi], p[64 * lo:64 * hi], 1) if hi > lo else b"\x00" * 64
            buf += pt + b"\x00" * 80
        return torch.frombuffer(buf, dtype=torch.uint8)

    def fold_fn(by_job, w, k):  # [job][rank][144]
        assert tuple(by_job.shape) == (k, w, 144)
        raw = bytes(by_job.numpy())
        out = bytearray()
        for i in range(k):
            acc = b"\x00" * 64
            for r in range(w):
                o = 144 * (i * w + r)
                acc = C.g1_add(acc, raw[o:o + 64])
            out += acc
        return torch.frombuffer(out, dtype=torch.uint8)

    out = ShardedMsmBatch(partials_fn, fold_fn).run(list(sizes))
    q.put((rank, bytes(out.numpy())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_msm_batch_one_collective_for_the_whole_batch(world):
    """K MSMs of different sizes (some smaller than the world: empty trailing shards), every one point-sharded, ONE
    all-gather of K x 144 bytes per rank, the gathered [rank][job] array transposed to [job][rank] before the folds: every
    rank ends with the K single-process results.  (The wiring `bench.py --gpus N` times: gpu_sharded_msm_batch.)"""
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as C

    sizes = (1, 2, 37, 500, 3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + world) % 2000
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
    exp = b"".join(C.msm_pippenger(C.sample_scalars(70 + i, n), C.sample_points(80 + i, n), 1) for i, n in enumerate(sizes))
    assert sorted(r for r, _ in got) == list(range(world))
    for _, out in got:
        assert out == exp


class _HostCtx:
    """CPU double of `Context` for the two entry points `gpu_sharded_msm_batch` calls: the "device pointers" are host
    addresses of CPU tensors (ctypes reads / writes them), the arithmetic is the C oracle's.  Like the device entry point
    it REFUSES an empty MSM (SNARKV_ERR_EMPTY) -- which is exactly what used to hang the peers in the all-gather."""

    def __init__(self):
        self.calls = []
        self.trace = []  # the order of stream-ordering calls and launches ("wait", "launch", "release")

    def wait_stream(self, stream=None):  # `snarkv_ctx_wait_stream`: nothing to order on the CPU, but the ORDER is checked
        self.trace.append("wait")

    def stream_wait(self, stream=None):  # `snarkv_stream_wait_ctx`
        self.trace.append("release")

    def msm_pippenger_many_partial_dev(self, ds, dp, counts, out_ptr, window_bits=0):
        import ctypes

        import coracle as C

        self.calls.append(list(counts))
        self.trace.append("launch")
        if any(c <= 0 for c in counts):
            raise RuntimeError("SNARKV_ERR_EMPTY")
        for i, (s, p, c) in enumerate(zip(ds, dp, counts)):
            pt = C.msm_pippenger(ctypes.string_at(s, 32 * c), ctypes.string_at(p, 64 * c), 1)
            ctypes.memmove(out_ptr + 144 * i, pt + bytes(80), 144)

    def fold_partials_many_dev(self, ptr, world, k, out_ptr):
        import ctypes

        import coracle as C

        self.trace.append("launch")
        raw = ctypes.string_at(ptr, 144 * world * k)
        for i in range(k):
            acc = bytes(64)
            for r in range(world):
                o = 144 * (i * world + r)
                acc = C.g1_add(acc, raw[o:o + 64])
            ctypes.memmove(out_ptr + 64 * i, acc, 64)


def _product_batch_worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    import coracle as C
    from snark_verifier_amd.distributed import gpu_sharded_msm_batch, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds, dp, counts = [], [], []
    for i, n in enumerate(sizes):  # this rank's shard of MSM i, as the weak-scaling caller holds it
        lo, hi = shard_range(n, rank, world)
        s, p = C.sample_scalars(70 + i, n), C.sample_points(80 + i, n)
        ds.append(torch.frombuffer(bytearray(s[32 * lo:32 * hi] or bytes(32)), dtype=torch.uint8))
        dp.append(torch.frombuffer(bytearray(p[64 * lo:64 * hi] or bytes(64)), dtype=torch.uint8))
        counts.append(hi - lo)
    ctx = _HostCtx()
    out = gpu_sharded_msm_batch(ctx, ds, dp, counts)
    # every launch is bracketed: the context waits for torch's stream before it, torch's stream for the context after it
    # (a context on a private stream races with torch's fills and with the collective otherwise: VERDICT r5)
    for i, ev in enumerate(ctx.trace):
        if ev == "launch":
            assert ctx.trace[i - 1] == "wait" and "release" in ctx.trace[i + 1:i + 2], ctx.trace
    q.put((rank, bytes(out.numpy()), counts, ctx.calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_product_batch_wiring_with_empty_local_shards(world):
    """`gpu_sharded_msm_batch` ITSELF (the function `bench.py --gpus N` times) at world > 1 with MSMs smaller than the
    world: the ranks whose shard of a job is empty must not hand that job to the device entry point (it fails with
    SNARKV_ERR_EMPTY on that rank only and the peers would block in the collective) -- its 144-byte slot
    stays the identity -- and every rank still ends with every job's single-process result."""
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as C

    sizes = (1, 2, 37, 500, 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + world) % 2000
    procs = [ctx.Process(target=_product_batch_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    exp = b"".join(C.msm_pippenger(C.sample_scalars(70 + i, n), C.sample_points(80 + i, n), 1) for i, n in enumerate(sizes))
    assert any(0 in counts for _, _, counts, _ in got)  # the case under test occurred
    for _, out, counts, calls in got:
        assert out == exp
        assert all(all(c > 0 for c in call) for call in calls)  # no empty job ever reached the device entry point
        assert sum(len(c) for c in calls) == sum(1 for c in counts if c > 0)


def test_shard_range_is_reference_chunking():
    from snark_verifier_amd.distributed import shard_range

    for n, w in [(10, 3), (16, 8), (5, 8), (1 << 24, 8)]:
        chunk = -(-n // w)
        cover = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            assert hi - lo <= chunk
            cover += list(range(lo, min(hi, lo + 3)))
        los = [shard_range(n, r, w)[0] for r in range(w)]
        assert los == sorted(los) and shard_range(n, w - 1, w)[1] == n


def _agg_worker(rank, world, port, n_proofs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random

    import torch.distributed as dist

    import bn254 as O
    import kzg as K
    import transcript as T
    from snark_verifier_amd.distributed import ShardedAggregation, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    secret = 0x5EC2E7
    rng = random.Random(99)  # every rank derives the same "proofs": one valid accumulator each
    accs = []
    for _ in range(n_proofs):
        a = rng.randrange(1, O.R)
        accs.append((O.g1_mul(O.G1_GEN, secret * a % O.R), O.g1_mul(O.G1_GEN, a)))

    def verify_fn(lo, hi):  # oracle double of the succinct verifier: the shard's accumulators
        return b"".join(O.g1_to_bytes(l) + O.g1_to_bytes(r) for l, r in accs[lo:hi])

    def combine_fn(allb):  # oracle double of KzgAs::create_proof + decide
        pairs = [(O.g1_from_bytes(allb[128 * i:128 * i + 64]), O.g1_from_bytes(allb[128 * i + 64:128 * i + 128]))
                 for i in range(len(allb) // 128)]
        t = T.EvmTranscript()
        for l, r in pairs:
            t.common_ec_point(l)
            t.common_ec_point(r)
        lhs, rhs = K.kzg_as_verify(pairs, t.squeeze_challenge())
        return O.g1_to_bytes(lhs) + O.g1_to_bytes(rhs), lhs == O.g1_mul(rhs, secret)

    acc, ok = ShardedAggregation(verify_fn, combine_fn).run(n_proofs)
    # a shard that REJECTS (an invalid proof: the normal failure) on ONE rank must come back as a reject on every
    # rank, after the collective -- not as an exception that leaves the peers blocked in the all-gather
    failing = n_proofs - 1  # the last proof: owned by the last non-empty rank

    def verify_fail(lo, hi):
        if lo <= failing < hi:
            raise RuntimeError("succinct verification failed on this shard")
        return verify_fn(lo, hi)

    def verify_none(lo, hi):
        return None if lo <= 0 < hi else verify_fn(lo, hi)

    rejected = [ShardedAggregation(verify_fail, combine_fn).run(n_proofs), ShardedAggregation(verify_none, combine_fn).run(n_proofs)]
    q.put((rank, acc, ok, shard_range(n_proofs, rank, world), rejected))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_proofs", [1, 2, 7])
def test_sharded_aggregation_two_ranks_agree(n_proofs):
    """Proof-sharded aggregation (distributed.py ShardedAggregation): uneven shards, an empty
    shard (n = 1), gather order = proof order, both ranks reach the same accumulator + verdict."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + n_proofs) % 2000
    procs = [ctx.Process(target=_agg_worker, args=(r, 2, port, n_proofs, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
    assert res[0][1] == res[1][1] and res[0][2] is True and res[1][2] is True
    assert res[0][3][1] == res[1][3][0]  # contiguous shards
    for r in res:  # the failing shard (raised / returned None) is a reject everywhere
        assert r[4] == [(None, False), (None, False)]
    # single-process reference: same fold over all accumulators in proof order
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import random

    import bn254 as O
    import kzg as K
    import transcript as T

    rng = random.Random(99)
    accs = []
    for _ in range(n_proofs):
        a = rng.randrange(1, O.R)
        accs.append((O.g1_mul(O.G1_GEN, 0x5EC2E7 * a % O.R), O.g1_mul(O.G1_GEN, a)))
    t = T.EvmTranscript()
    for l, r in accs:
        t.common_ec_point(l)
        t.common_ec_point(r)
    lhs, rhs = K.kzg_as_verify(accs, t.squeeze_challenge())
    assert res[0][1] == O.g1_to_bytes(lhs) + O.g1_to_bytes(rhs)


# ---- bucket-sharded variant ("bucket-sum allreduce") ----------------------------
_BC, _BW = 8, 32  # doubles: unsigned 8-bit windows over the 256-bit scalar, 255 buckets each


def _bucket_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist

    import coracle as C
    from snark_verifier_amd.distributed import BucketShardedMsm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, p = C.sample_scalars(7, n), C.sample_points(8, n)
    B = (1 << _BC) - 1
    Z = b"\x00" * 64

    def fill_fn(lo, hi):  # oracle double of the bucket fill: affine 64-byte buckets, zero = identity
        grid = [Z] * (_BW * B)
        for i in range(lo, hi):
            k = int.from_bytes(s[32 * i:32 * i + 32], "little")
            for w in range(_BW):
                d = (k >> (_BC * w)) & B
                if d:
                    grid[w * B + d - 1] = C.g1_add(grid[w * B + d - 1], p[64 * i:64 * i + 64])
        return torch.frombuffer(bytearray(b"".join(grid)), dtype=torch.uint8)

    def add_fn(dst, src):
        a, b = bytes(dst.numpy()), bytes(src.numpy())
        r = b"".join(C.g1_add(a[o:o + 64], b[o:o + 64]) for o in range(0, len(a), 64))
        dst.copy_(torch.frombuffer(bytearray(r), dtype=torch.uint8))

    def reduce_fn(buckets, w0, wcount):
        raw = bytes(buckets.numpy())
        acc = Z
        for w in range(wcount):
            run, tot = Z, Z
            for b in range(B - 1, -1, -1):  # running-sum trick, msm.rs:298-302
                run = C.g1_add(run, raw[(w * B + b) * 64:(w * B + b) * 64 + 64])
                tot = C.g1_add(tot, run)
            acc = C.g1_add(acc, C.g1_mul(tot, (1 << (_BC * (w0 + w))).to_bytes(32, "little")))
        return torch.frombuffer(bytearray(acc + b"\x00" * 80), dtype=torch.uint8)

    def fold_fn(gathered, w):
        acc = Z
        raw = bytes(gathered.numpy())
        for k in range(w):
            acc = C.g1_add(acc, raw[144 * k:144 * k + 64])
        return torch.frombuffer(bytearray(acc), dtype=torch.uint8)

    out = BucketShardedMsm(_BW, B, fill_fn, add_fn, reduce_fn, fold_fn, bucket_bytes=64).run(n)
    q.put((rank, bytes(out.numpy())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(1, 2), (37, 2), (37, 3)])
def test_bucket_sharded_msm_ranks_agree_with_single_process(n, world):
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle as C

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + 7 * n + world) % 2000
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    expected = C.msm_pippenger(C.sample_scalars(7, n), C.sample_points(8, n), 1)
    for _, out in res:
        assert out == expected


# ---- sharded IPA decide ------------------------------------------------------------------------
def _ipa_worker(rank, world, port, k, good, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import random

    import torch
    import torch.distributed as dist

    import bn254 as O
    import coracle as C
    import ipa as I
    from snark_verifier_amd.distributed import ShardedIpaDecide

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rnd = random.Random(31)
    xi = [rnd.randrange(O.R) for _ in range(k)]
    g = C.sample_points(5, 1 << k)
    h = I.h_coeffs(xi, 1)
    u = C.msm_pippenger(b"".join(O.fe_to_bytes(c) for c in h), g, 1)
    if not good:
        u = C.g1_add(u, g[:64])

    def partial_fn(lo, hi, xi_):  # oracle double of the device partial: affine bytes padded to 144
        pt = C.msm_pippenger(b"".join(O.fe_to_bytes(c) for c in h[lo:hi]), g[64 * lo:64 * hi], 1)
        return torch.frombuffer(bytearray(pt + bytes(80)), dtype=torch.uint8)

    def fold_fn(gathered, w):
        acc, raw = bytes(64), bytes(gathered.numpy())
        for j in range(w):
            acc = C.g1_add(acc, raw[144 * j:144 * j + 64])
        return torch.frombuffer(bytearray(acc), dtype=torch.uint8)

    q.put((rank, ShardedIpaDecide(k, partial_fn, fold_fn).run(xi, u)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k,world,good", [(5, 2, True), (5, 2, False), (3, 3, True), (1, 3, True)])
def test_sharded_ipa_decide_ranks_agree(k, world, good):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() + 13 * k + world + good) % 2000
    procs = [ctx.Process(target=_ipa_worker, args=(r, world, port, k, good, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert [v for _, v in res] == [good] * world
