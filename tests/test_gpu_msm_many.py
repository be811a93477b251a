"""GPU parity of the batch entry point `snarkv_g1_msm_pippenger_many_dev` (include/snarkv_amd.h): MANY independent
`util::msm::multi_scalar_multiplication` calls (reference snark-verifier/src/util/msm.rs:308-343) in one call, pipelined
by the library (csrc/capi.hip launch_msm_pippenger_many).  Every job must give the bytes of the single-MSM entry point
and of the C oracle: uniform batches (one batched tail over grids laid end to end), ragged batches (per-job tails),
several rounds, the projective-partial form, the fall-backs, the error codes."""
import os

import pytest

import bn254 as O
import coracle as C

pytestmark = pytest.mark.gpu


def _upload(torch, jobs):
    ds = [torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda() for s, _ in jobs]
    dp = [torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda() for _, p in jobs]
    return ds, dp


def _many(ctx, torch, jobs, partial=False, window_bits=0):
    import snark_verifier_amd as sv

    ds, dp = _upload(torch, jobs)
    counts = [len(s) // 32 for s, _ in jobs]
    stride = sv.G1_PARTIAL_BYTES if partial else 64
    out = torch.zeros(stride * max(1, len(jobs)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    fn = ctx.msm_pippenger_many_partial_dev if partial else ctx.msm_pippenger_many_dev
    fn([t.data_ptr() for t in ds], [t.data_ptr() for t in dp], counts, out.data_ptr(), window_bits)
    ctx.sync()
    raw = bytes(out.cpu().numpy())
    return [raw[stride * i:stride * (i + 1)] for i in range(len(jobs))]


def _jobs(sizes, seed):
    return [(C.sample_scalars(seed + 2 * i, n), C.sample_points(seed + 2 * i + 1, n)) for i, n in enumerate(sizes)]


@pytest.mark.parametrize("sizes", [[3000] * 6, [1] * 3, [4097] * 2, [65536] * 5,
                                   [1, 2, 77, 4096, 65536, 100_000, 3],  # ragged: window sizes differ from job to job
                                   [50_000, 50_001, 49_999, 50_000]])     # ragged sizes, one window size
def test_every_job_matches_the_oracle(gpu_ctx, sizes):
    import torch

    jobs = _jobs(sizes, 0x4D00 + len(sizes))
    got = _many(gpu_ctx, torch, jobs)
    for i, (s, p) in enumerate(jobs):
        assert got[i] == C.msm_pippenger(s, p, 8), (sizes, i)
        assert got[i] == gpu_ctx.msm_pippenger(s, p), (sizes, i)
    assert _many(gpu_ctx, torch, jobs) == got  # the job contexts are reused: same bytes again


def test_more_jobs_than_a_round_holds(gpu_ctx, monkeypatch):
    """70 jobs > SNARKV_MANY_MAX_JOBS: successive rounds; and rounds of 3 forced through the tuning knob."""
    import torch

    jobs = _jobs([2000 + 7 * i for i in range(70)], 0x4E00)
    exp = [C.msm_pippenger(s, p, 8) for s, p in jobs]
    assert _many(gpu_ctx, torch, jobs) == exp
    monkeypatch.setenv("SNARKV_MANY_JOBS", "3")
    assert _many(gpu_ctx, torch, jobs[:10]) == exp[:10]
    monkeypatch.setenv("SNARKV_MANY_JOBS", "1")
    assert _many(gpu_ctx, torch, jobs[:4]) == exp[:4]


def test_identity_points_zero_scalars_and_repeated_points(gpu_ctx):
    import torch

    n = 3000
    p = C.sample_points(5, n)
    s = C.sample_scalars(6, n)
    ident = bytes(64) * n
    zeros = bytes(32) * n
    same_point = p[:64] * n  # every term on one point: buckets collapse, the careful adders run
    jobs = [(s, ident), (zeros, p), (s, same_point), (s, p)]
    got = _many(gpu_ctx, torch, jobs)
    assert got[0] == bytes(64) and got[1] == bytes(64)
    assert got[2] == C.msm_pippenger(s, same_point, 4)
    assert got[3] == C.msm_pippenger(s, p, 4)


def test_partial_form_folds_to_the_same_point(gpu_ctx):
    """multi-GPU use: every rank's shard as projective partials, folded per MSM"""
    import torch

    import snark_verifier_amd as sv

    whole = _jobs([40_000] * 3, 0x4F00)
    halves = [[(s[:32 * 15_000], p[:64 * 15_000]) for s, p in whole], [(s[32 * 15_000:], p[64 * 15_000:]) for s, p in whole]]
    parts = [_many(gpu_ctx, torch, h, partial=True) for h in halves]
    for i, (s, p) in enumerate(whole):
        buf = torch.frombuffer(bytearray(parts[0][i] + parts[1][i]), dtype=torch.uint8).cuda()
        out = torch.zeros(64, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        gpu_ctx.fold_partials_dev(buf.data_ptr(), 2, out.data_ptr())
        gpu_ctx.sync()
        assert bytes(out.cpu().numpy()) == C.msm_pippenger(s, p, 8)
    assert len(parts[0][0]) == sv.G1_PARTIAL_BYTES
    # all the folds in one launch: [job][rank] layout
    buf = torch.frombuffer(bytearray(b"".join(parts[0][i] + parts[1][i] for i in range(3))), dtype=torch.uint8).cuda()
    out = torch.zeros(64 * 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.fold_partials_many_dev(buf.data_ptr(), 2, 3, out.data_ptr())
    gpu_ctx.sync()
    got = bytes(out.cpu().numpy())
    assert [got[64 * i:64 * i + 64] for i in range(3)] == [C.msm_pippenger(s, p, 8) for s, p in whole]


def test_window_bits_and_fallbacks(gpu_ctx, monkeypatch):
    import torch

    jobs = _jobs([20_000] * 3, 0x5000)
    exp = [C.msm_pippenger(s, p, 8) for s, p in jobs]
    for c in (9, 12, 15):
        assert _many(gpu_ctx, torch, jobs, window_bits=c) == exp, c
    assert _many(gpu_ctx, torch, jobs[:1]) == exp[:1]  # one job: the single-call path
    monkeypatch.setenv("SNARKV_MANY_MODE", "0")  # A/B knob: one MSM after the other
    assert _many(gpu_ctx, torch, jobs) == exp


def test_error_codes_and_empty_batch(gpu_ctx):
    import torch

    import snark_verifier_amd as sv

    assert _many(gpu_ctx, torch, []) == []
    s, p = C.sample_scalars(1, 10), C.sample_points(2, 10)
    ds, dp = _upload(torch, [(s, p), (s, p)])
    out = torch.zeros(128, dtype=torch.uint8, device="cuda")
    with pytest.raises(sv.SnarkvError) as e:  # an empty MSM inside a batch: the reference panics (msm.rs:265)
        gpu_ctx.msm_pippenger_many_dev([t.data_ptr() for t in ds], [t.data_ptr() for t in dp], [10, 0], out.data_ptr())
    assert e.value.code == -1
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_pippenger_many_dev([ds[0].data_ptr(), 0], [t.data_ptr() for t in dp], [10, 10], out.data_ptr())
    assert e.value.code == -5


def test_the_bench_shape_2p20_times_six_against_single_calls():
    """six 2^20-point MSMs (the shape bench.py submits, shortened): batch == single calls == C oracle on job 0"""
    import torch

    import snark_verifier_amd as sv

    ctx = sv.Context(0)
    n, k = 1 << 20, 6
    ds = [torch.empty(32 * n, dtype=torch.uint8, device="cuda") for _ in range(k)]
    dp = [torch.empty(64 * n, dtype=torch.uint8, device="cuda") for _ in range(k)]
    torch.cuda.synchronize()
    for i in range(k):
        ctx.sample_scalars_dev(0x5EED0001, n, ds[i].data_ptr(), first=i * n)
        ctx.sample_points_dev(0x5EED0002, n, dp[i].data_ptr(), first=i * n)
    out = torch.zeros(64 * k, dtype=torch.uint8, device="cuda")
    single = torch.zeros(64 * k, dtype=torch.uint8, device="cuda")
    ctx.sync()
    ctx.msm_pippenger_many_dev([t.data_ptr() for t in ds], [t.data_ptr() for t in dp], [n] * k, out.data_ptr())
    ctx.sync()
    for i in range(k):
        ctx.msm_pippenger_dev(ds[i].data_ptr(), dp[i].data_ptr(), n, single.data_ptr() + 64 * i)
    ctx.sync()
    got = bytes(out.cpu().numpy())
    assert got == bytes(single.cpu().numpy())
    assert got[:64] == C.msm_pippenger(bytes(ds[0].cpu().numpy()), bytes(dp[0].cpu().numpy()), os.cpu_count() or 1)
    ctx.set_stage_timing(True)
    ctx.msm_pippenger_many_dev([t.data_ptr() for t in ds], [t.data_ptr() for t in dp], [n] * k, out.data_ptr())
    st = ctx.get_stage_timing()
    assert st["total"] > 0 and st["bucket_accumulate"] > 0
    assert bytes(out.cpu().numpy()) == got
    ctx.close()


def test_host_resident_batch(gpu_ctx, monkeypatch):
    """`snarkv_g1_msm_pippenger_many` (host pointers: the uploads run on a copy stream under the kernels of earlier jobs):
    every job = the oracle, from pageable bytes and from the context's pinned buffers, uniform / ragged / more jobs than a
    round / the one-after-the-other fall-back; validation and the error codes of the single-call entry point."""
    import ctypes

    import snark_verifier_amd as sv

    for sizes in ([30_000] * 5, [1, 77, 4096, 65_536, 3], [2000 + 11 * i for i in range(70)]):
        jobs = _jobs(sizes, 0x5600 + len(sizes))
        exp = [C.msm_pippenger(s, p, 8) for s, p in jobs]
        assert gpu_ctx.msm_pippenger_many_host([s for s, _ in jobs], [p for _, p in jobs], sizes) == exp, sizes[:3]
    # pinned sources (snarkv_ctx_host_buffer), passed by address
    sizes = [50_000, 20_000, 50_000]
    jobs = _jobs(sizes, 0x5700)
    hs, hp = gpu_ctx.host_buffer(2, 32 * sum(sizes)), gpu_ctx.host_buffer(3, 64 * sum(sizes))
    ps, pp, off = [], [], 0
    for (s, p), n in zip(jobs, sizes):
        ctypes.memmove(ctypes.addressof(hs) + 32 * off, s, 32 * n)
        ctypes.memmove(ctypes.addressof(hp) + 64 * off, p, 64 * n)
        ps.append(ctypes.addressof(hs) + 32 * off), pp.append(ctypes.addressof(hp) + 64 * off)
        off += n
    exp = [C.msm_pippenger(s, p, 8) for s, p in jobs]
    assert gpu_ctx.msm_pippenger_many_host(ps, pp, sizes) == exp
    assert gpu_ctx.msm_pippenger_many_host(ps, pp, sizes, sv.SNARKV_FLAG_VALIDATE) == exp
    monkeypatch.setenv("SNARKV_MANY_MODE", "0")  # one MSM after the other: each waits for its own upload
    assert gpu_ctx.msm_pippenger_many_host(ps, pp, sizes) == exp
    monkeypatch.delenv("SNARKV_MANY_MODE")
    bad = bytearray(jobs[1][1])
    bad[64 * 5] ^= 1
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_pippenger_many_host([s for s, _ in jobs], [jobs[0][1], bytes(bad), jobs[2][1]], sizes, sv.SNARKV_FLAG_VALIDATE)
    assert e.value.code == -3
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_pippenger_many_host([jobs[0][0], b"\x00" * 32], [jobs[0][1], b"\x00" * 64], [sizes[0], 0])
    assert e.value.code == -1
    assert gpu_ctx.msm_pippenger_many_host([], [], []) == []
