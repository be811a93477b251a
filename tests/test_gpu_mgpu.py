"""GPU: the single-process multi-GPU C ABI (include/snarkv_amd.h `snarkv_mgpu_*`, csrc/mgpu.hip; SURVEY.md 8b
`*_multi_gpu`, 8e) on a 1-GPU box: every rank is a context of its own on device 0, so the sharding (the reference's
`chunk = ceil(n / ranks)`, util/msm.rs:311-336), both combine variants (point-sharded / bucket-sharded "bucket-sum
allreduce"), the peer copies, events and folds of an 8-rank run are all executed -- against the C oracle, bit for bit.
BASELINE configs 4 and 5 in their single-process form."""
import os

import pytest

import bn254 as O
import coracle as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("variant", [0, 1])
def test_mgpu_msm_host_buffers_vs_oracle(world, variant):
    import snark_verifier_amd as sv

    mg = sv.MultiGpu([0] * world)
    assert mg.world == world
    for n in (1, 5, 9, 777, 5000, (1 << 16) + 3):  # n < world: trailing ranks hold the identity
        s, p = C.sample_scalars(0x40 + n, n), C.sample_points(0x41 + n, n)
        assert mg.msm_pippenger(s, p, variant) == C.msm_pippenger(s, p, 8), (world, variant, n)
    # duplicate / opposite / identity bases that end up in DIFFERENT shards and meet again in the exchange / fold
    n = 4096
    base = C.sample_points(0x77, n // 4)
    p = bytearray(base * 4)
    neg = O.g1_to_bytes(O.g1_neg(O.g1_from_bytes(bytes(p[:64]))))
    p[64 * 2048:64 * 2049] = neg
    p[64 * 7:64 * 8] = bytes(64)
    s = bytearray(C.sample_scalars(0x78, n))
    s[32 * 2048:32 * 2049] = s[:32]  # s_0 * P + s_0 * (-P) cancels across shards
    assert mg.msm_pippenger(bytes(s), bytes(p), variant) == C.msm_pippenger(bytes(s), bytes(p), 8)
    mg.close()


def test_mgpu_shard_rule_and_errors():
    import snark_verifier_amd as sv
    from snark_verifier_amd.distributed import shard_range

    mg = sv.MultiGpu([0, 0, 0])
    for n in (1, 2, 3, 10, 1000):
        assert [mg.shard(n, r) for r in range(3)] == [shard_range(n, r, 3) for r in range(3)]
    with pytest.raises(sv.SnarkvError) as e:
        mg.msm_pippenger(b"", b"")
    assert e.value.code == -1  # empty MSM: the reference panics (msm.rs:265)
    with pytest.raises(sv.SnarkvError) as e:
        mg.msm_pippenger(bytes(64), bytes(64))
    assert e.value.code == -2
    with pytest.raises(sv.SnarkvError):
        sv.MultiGpu([])
    with pytest.raises(sv.SnarkvError):
        sv.MultiGpu([99])  # no such device: loud failure, no fallback
    mg.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_mgpu_config4_shape_device_resident_shards(variant):
    """BASELINE config 4 in shape: 8 ranks, shards generated in place on each rank's device (disjoint index ranges of
    the bench's seeded streams), 2^21 points in total here (2^18 per rank) so the threaded C oracle checks the bytes;
    the 2^24 total of the config is the same code with bigger shards (bench.py --total-log2n 24)."""
    import torch

    import snark_verifier_amd as sv

    world, per = 8, 1 << 18
    mg = sv.MultiGpu([0] * world)
    ds, dp = [], []
    for r in range(world):
        c = mg.rank_context(r)
        s = torch.empty(32 * per, dtype=torch.uint8, device="cuda")
        p = torch.empty(64 * per, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        c.sample_scalars_dev(0x5EED0001, per, s.data_ptr(), first=r * per)
        c.sample_points_dev(0x5EED0002, per, p.data_ptr(), first=r * per)
        c.sync()
        ds.append(s)
        dp.append(p)
    got = mg.msm_pippenger_dev([t.data_ptr() for t in ds], [t.data_ptr() for t in dp], [per] * world, 0, variant)
    s = b"".join(bytes(t.cpu().numpy()) for t in ds)
    p = b"".join(bytes(t.cpu().numpy()) for t in dp)
    assert s[:32 * 16] == C.sample_scalars(0x5EED0001, 16)
    assert got == C.msm_pippenger(s, p, os.cpu_count() or 1)
    # uneven shards incl. an empty one
    counts = [per, 0, 1000, per, 7, per // 2, 1, per]
    s2 = b"".join(bytes(ds[r][:32 * counts[r]].cpu().numpy()) for r in range(world))
    p2 = b"".join(bytes(dp[r][:64 * counts[r]].cpu().numpy()) for r in range(world))
    got = mg.msm_pippenger_dev([t.data_ptr() for t in ds], [t.data_ptr() for t in dp], counts, 0, variant)
    assert got == C.msm_pippenger(s2, p2, os.cpu_count() or 1)
    mg.close()


@pytest.mark.parametrize("world", [1, 3, 8])
def test_mgpu_decide_all_1024_sharded(world):
    """BASELINE config 5's decider in its single-process multi-GPU form: 1 024 distinct accumulators of the committed
    fixture, k invalid ones at known indices, sharded over `world` ranks: verdicts in order, identical for any rank count."""
    import snark_verifier_amd as sv
    from snark_verifier_amd import host_api as H

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = H.read_fixture(os.path.join(root, "tests", "golden", "bench_plonk_gwc19_evm_1024.bin"))
    dk, accs = fx["dk"], bytearray(fx["accs"])
    bad = [0, 127, 128, 341, 342, 682, 683, 1023]  # around the shard boundaries of 3 and 8 ranks
    g = O.g1_to_bytes(O.G1_GEN)
    for i in bad:
        accs[128 * i:128 * i + 64] = C.g1_add(bytes(accs[128 * i:128 * i + 64]), g)
    mg = sv.MultiGpu([0] * world)
    allok, oks = mg.decide_batch(dk[:64], dk[64:192], dk[192:320], bytes(accs))
    assert not allok and oks == [i not in bad for i in range(1024)]
    allok, oks = mg.decide_batch(dk[:64], dk[64:192], dk[192:320], fx["accs"])
    assert allok and all(oks)
    for m in (1, 2, 5):  # fewer accumulators than ranks
        allok, oks = mg.decide_batch(dk[:64], dk[64:192], dk[192:320], bytes(accs[:128 * m]))
        assert oks == [i not in bad for i in range(m)]
    assert mg.decide_batch(dk[:64], dk[64:192], dk[192:320], b"") == (True, [])
    mg.close()


def test_mgpu_large_shards_take_the_chunk_pipeline(gpu_ctx, monkeypatch):
    """Shards of more than 3 x 2^20 points run as the chunk pipeline over shared bucket grids INSIDE each rank
    (capi.hip launch_msm_pippenger_auto): two ranks x (3 x 2^20 + 5) points == one single-launch MSM over all of them."""
    import torch

    import snark_verifier_amd as sv

    per = 3 * (1 << 20) + 5
    n = 2 * per
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.sample_scalars_dev(0x6A, n, ds.data_ptr())
    gpu_ctx.sample_points_dev(0x6B, n, dp.data_ptr())
    monkeypatch.setenv("SNARKV_PIP_SPLIT", "0")
    gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr())
    gpu_ctx.sync()
    single = bytes(out.cpu().numpy())
    monkeypatch.setenv("SNARKV_PIP_SPLIT", "1")
    mg = sv.MultiGpu([0, 0])
    got = mg.msm_pippenger_dev([ds.data_ptr(), ds.data_ptr() + 32 * per], [dp.data_ptr(), dp.data_ptr() + 64 * per], [per, per])
    assert got == single != bytes(64)
    mg.close()


def test_mgpu_every_rank_holds_the_result_and_peer_status():
    """All-reduce semantics: after an MSM every rank's device holds the 64-byte result, not only rank 0's;
    peer-access status is reported, not swallowed (on this 1-GPU box no pair of DISTINCT devices exists: all zero)."""
    import ctypes

    import torch

    import snark_verifier_amd as sv

    for variant in (0, 1):
        mg = sv.MultiGpu([0] * 4)
        assert mg.peer_access() == (0, 0, 0)
        n = 3000
        s, p = C.sample_scalars(0x61, n), C.sample_points(0x62, n)
        exp = C.msm_pippenger(s, p, 8)
        assert mg.msm_pippenger(s, p, variant) == exp
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        for r in range(4):
            src = mg.result_dev(r)
            assert src
            tmp = torch.empty(64, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            assert hip.hipMemcpy(ctypes.c_void_p(tmp.data_ptr()), ctypes.c_void_p(src), 64, 3) == 0  # hipMemcpyDeviceToDevice
            assert bytes(tmp.cpu().numpy()) == exp, (variant, r)
        mg.close()


def test_mgpu_rccl_transport_world_1_and_its_refusals():
    """The RCCL transport of the C ABI (`ncclCommInitAll` + a grouped `ncclAllGather` of the 144-byte partials -- what
    BASELINE's north_star names): on a 1-GPU box it can run with ONE rank (a communicator of one device); a device listed
    twice is refused loudly (RCCL cannot build that communicator) and the handle keeps working on peer copies."""
    import snark_verifier_amd as sv

    n = 5000
    s, p = C.sample_scalars(0x63, n), C.sample_points(0x64, n)
    exp = C.msm_pippenger(s, p, 8)
    mg = sv.MultiGpu([0])
    mg.set_transport(sv.MultiGpu.RCCL)
    assert mg.msm_pippenger(s, p, 0) == exp
    assert mg.msm_pippenger(s, p, 1) == exp
    mg.set_transport(sv.MultiGpu.PEER_COPY)
    assert mg.msm_pippenger(s, p, 0) == exp
    mg.close()
    mg = sv.MultiGpu([0, 0])
    with pytest.raises(sv.SnarkvError) as e:
        mg.set_transport(sv.MultiGpu.RCCL)
    assert "distinct devices" in str(e.value)
    assert mg.msm_pippenger(s, p, 0) == exp  # still on peer copies
    mg.close()


def _rank_shards(mg, world, jobs, sizes):
    """rank g's shard of job j: the reference's ceil chunking of job j's `sizes[j]` points, sampled in place on rank g's
    device from disjoint ranges of one seeded stream per job; returns (tensors, pointers, counts, whole-job bytes)"""
    import torch

    keep, ds, dp, cn = [], [], [], []
    whole = [(C.sample_scalars(0x900 + j, sizes[j]), C.sample_points(0xA00 + j, sizes[j])) for j in range(jobs)]
    for g in range(world):
        c = mg.rank_context(g)
        rs, rp, rc = [], [], []
        for j in range(jobs):
            lo, hi = mg.shard(sizes[j], g)
            m = hi - lo
            if m == 0:
                rs.append(None), rp.append(None), rc.append(0)
                continue
            s = torch.empty(32 * m, dtype=torch.uint8, device="cuda")
            p = torch.empty(64 * m, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            c.sample_scalars_dev(0x900 + j, m, s.data_ptr(), first=lo)
            c.sample_points_dev(0xA00 + j, m, p.data_ptr(), first=lo)
            c.sync()
            keep += [s, p]
            rs.append(s.data_ptr()), rp.append(p.data_ptr()), rc.append(m)
        ds.append(rs), dp.append(rp), cn.append(rc)
    return keep, ds, dp, cn, whole


@pytest.mark.parametrize("world", [1, 2, 8])
def test_mgpu_batch_of_msms_one_exchange_vs_oracle(world):
    """`snarkv_g1_msm_pippenger_many_mgpu_dev` (the bench's `single_process_mgpu` leg): K jobs, each sharded over the
    ranks, ONE exchange of K x 144 B per rank, transposed and folded per job -- job by job the bytes of the C oracle on
    the whole job, on every rank's device too; uniform sizes take the batch pipeline on every rank."""
    import ctypes

    import torch

    import snark_verifier_amd as sv

    mg = sv.MultiGpu([0] * world)
    jobs = 5
    sizes = [1 << 14] * jobs
    keep, ds, dp, cn, whole = _rank_shards(mg, world, jobs, sizes)
    exp = [C.msm_pippenger(s, p, 8) for s, p in whole]
    got = mg.msm_pippenger_many_dev(ds, dp, cn)
    assert got == exp and len(set(got)) == jobs
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    for r in range(world):
        tmp = torch.empty(64 * jobs, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        assert hip.hipMemcpy(ctypes.c_void_p(tmp.data_ptr()), ctypes.c_void_p(mg.results_many_dev(r)), 64 * jobs, 3) == 0
        assert bytes(tmp.cpu().numpy()) == b"".join(exp), r
    # a second batch on the same handle, more jobs than before (the buffers grow), ragged sizes (every job its own tail)
    jobs = 37
    sizes = [1 + 97 * j for j in range(jobs)]  # job 0 has ONE point: with world > 1 the trailing ranks hold empty shards
    keep, ds, dp, cn, whole = _rank_shards(mg, world, jobs, sizes)
    assert mg.msm_pippenger_many_dev(ds, dp, cn) == [C.msm_pippenger(s, p, 8) for s, p in whole]
    mg.close()


def test_mgpu_batch_rccl_transport_world_1_and_errors():
    import snark_verifier_amd as sv

    mg = sv.MultiGpu([0])
    mg.set_transport(sv.MultiGpu.RCCL)
    keep, ds, dp, cn, whole = _rank_shards(mg, 1, 3, [5000, 5000, 5000])
    assert mg.msm_pippenger_many_dev(ds, dp, cn) == [C.msm_pippenger(s, p, 8) for s, p in whole]
    with pytest.raises(sv.SnarkvError) as e:  # a job with no point on any rank: the reference panics (msm.rs:265)
        mg.msm_pippenger_many_dev([[ds[0][0], None]], [[dp[0][0], None]], [[5000, 0]])
    assert e.value.code == -1
    mg.close()
