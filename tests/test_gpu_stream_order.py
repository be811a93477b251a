"""Stream ordering at the boundary (SURVEY.md 8b "Threading"): a context on its PRIVATE
non-blocking stream, inputs filled and outputs consumed by torch on torch's stream, no host synchronisation in between --
`snarkv_ctx_wait_stream` / `snarkv_stream_wait_ctx` (include/snarkv_amd.h) are the only ordering.  Every iteration
changes the inputs right before the call, behind a deliberately slow torch-side fill, so a launch that does not wait for
torch's stream reads the PREVIOUS iteration's inputs (or has its output overwritten by the late fill) and the bytes
differ from the oracle's."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import coracle as C  # noqa: E402  (the CPU checker)

pytestmark = pytest.mark.gpu

ITERS = 200


def _sets(n, k, seed):
    import torch

    out = []
    for i in range(k):
        s, p = C.sample_scalars(seed + 2 * i, n), C.sample_points(seed + 2 * i + 1, n)
        out.append((torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda(), torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda(),
                    C.msm_pippenger(s, p, 2)))
    torch.cuda.synchronize()
    return out


def test_the_primitives_order_a_private_stream_against_torch():
    """The C ABI pair alone: torch fills the inputs and a poisoned output, the context (private stream) waits, computes,
    and torch's stream waits for it before copying the output away -- ITERS times without one host synchronisation."""
    import torch

    import snark_verifier_amd as sv

    ctx = sv.Context(0, ordered=False)  # private non-blocking stream, NO automatic ordering: the explicit calls below are the only one
    assert ctx.stream != 0 and ctx.stream != torch.cuda.current_stream().cuda_stream
    n = 2048
    sets = _sets(n, 3, 0x600)
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    ballast = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    kept = torch.zeros(ITERS, 64, dtype=torch.uint8, device="cuda")
    for it in range(ITERS):
        s, p, _ = sets[it % 3]
        ballast.fill_(it & 0xFF)  # ~0.1 ms of torch-stream work in front of the fills the context depends on
        ds.copy_(s)
        dp.copy_(p)
        out = torch.full((64,), 0xAB, dtype=torch.uint8, device="cuda")  # a late fill would overwrite the result
        ctx.wait_stream()
        ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr())
        ctx.stream_wait()
        kept[it] = out  # torch's stream: must see the context's write; `out` is freed and reused next iteration
    torch.cuda.synchronize()
    got = bytes(kept.cpu().numpy())
    for it in range(ITERS):
        assert got[64 * it:64 * it + 64] == sets[it % 3][2], it
    ctx.close()


def test_a_context_on_the_callers_stream_is_ordered_already():
    """`hip_stream` == the context's own stream: both calls are no-ops that succeed (also with the NULL stream handle)."""
    import torch

    import snark_verifier_amd as sv

    side = torch.cuda.Stream()
    ctx = sv.Context(0, stream=side.cuda_stream)
    assert ctx.stream == side.cuda_stream
    with torch.cuda.stream(side):
        ctx.wait_stream()
        ctx.stream_wait()
    ctx.wait_stream(side)
    ctx.stream_wait(side.cuda_stream)
    ctx.wait_stream(0)  # the legacy default stream
    ctx.stream_wait(0)
    ctx.sync()
    ctx.close()


@pytest.mark.parametrize("wiring", ["sharded", "bucket_sharded", "batch"])
def test_sharded_wiring_with_a_private_stream_and_slow_torch_fills(wiring):
    """`distributed.gpu_sharded_msm` / `gpu_bucket_sharded_msm` / `gpu_sharded_msm_batch` (the N > 1 product wiring,
    here at world 1) ITERS times each with a context on a private stream, the inputs rewritten by torch right before
    every call behind a large fill: equal to the oracle every time, with no host synchronisation inside the helpers."""
    import torch

    import snark_verifier_amd as sv
    from snark_verifier_amd import distributed as D

    ctx = sv.Context(0, ordered=False)  # the helpers' own wait_stream / stream_wait calls are the only ordering
    n = 1500
    sets = _sets(n, 3, 0x700)
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    ds2 = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp2 = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    ballast = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    results = []
    for it in range(ITERS):
        s, p, _ = sets[it % 3]
        ballast.fill_(it & 0xFF)
        ds.copy_(s)
        dp.copy_(p)
        if wiring == "sharded":
            results.append(D.gpu_sharded_msm(ctx, ds, dp, n))
        elif wiring == "bucket_sharded":
            results.append(D.gpu_bucket_sharded_msm(ctx, ds, dp, n))
        else:
            s2, p2, _ = sets[(it + 1) % 3]
            ds2.copy_(s2)
            dp2.copy_(p2)
            results.append(D.gpu_sharded_msm_batch(ctx, [ds, ds2, ds], [dp, dp2, dp], [n, n, 0]))  # an empty shard rides along
    torch.cuda.synchronize()
    for it, r in enumerate(results):
        got = bytes(r.cpu().numpy())
        if wiring == "batch":
            assert got == sets[it % 3][2] + sets[(it + 1) % 3][2] + bytes(64), it
        else:
            assert got == sets[it % 3][2], it
    ctx.close()


def test_the_per_rank_step_with_poisoned_partials():
    """The pattern of `test_empty_shard_contributes_the_identity` (the round-5 red test), ITERS times: a 0xAB-filled
    partial handed to `gpu_msm_partial` on a private-stream context and copied away by torch at once."""
    import torch

    import snark_verifier_amd as sv
    from snark_verifier_amd.distributed import gpu_msm_partial, shard_range

    ctx = sv.Context(0, ordered=False)
    n, world = 5, 8
    s, p = C.sample_scalars(0x91, n), C.sample_points(0x92, n)
    want = C.msm_pippenger(s, p, 1)
    ds = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
    dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
    outs = torch.zeros(ITERS, 64, dtype=torch.uint8, device="cuda")
    for it in range(ITERS):
        gathered = torch.zeros(world, sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
        for r in range(world):
            lo, hi = shard_range(n, r, world)
            part = torch.full((sv.G1_PARTIAL_BYTES,), 0xAB, dtype=torch.uint8, device="cuda")
            gpu_msm_partial(ctx, part, ds[32 * lo:], dp[64 * lo:], hi - lo)
            gathered[r] = part
        ctx.wait_stream()
        ctx.fold_partials_dev(gathered.data_ptr(), world, outs[it].data_ptr())
        ctx.stream_wait()
    torch.cuda.synchronize()
    got = bytes(outs.cpu().numpy())
    assert all(got[64 * it:64 * it + 64] == want for it in range(ITERS))
    ctx.close()


def test_a_private_stream_context_is_ordered_by_default():
    """`sv.Context(0)` (no stream passed) brackets every `*_dev` method with the two calls itself: the loop of the first
    test without any explicit ordering; `ordered=False` / an explicit stream leave the ordering to the caller."""
    import torch

    import snark_verifier_amd as sv

    ctx = sv.Context(0)
    assert ctx.ordered is True and sv.Context(0, ordered=False).ordered is False
    assert sv.Context(0, stream=torch.cuda.Stream().cuda_stream).ordered is False
    n = 2048
    sets = _sets(n, 3, 0x600)
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    ballast = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    kept = torch.zeros(ITERS, 64, dtype=torch.uint8, device="cuda")
    for it in range(ITERS):
        s, p, _ = sets[it % 3]
        ballast.fill_(it & 0xFF)
        ds.copy_(s)
        dp.copy_(p)
        out = torch.full((64,), 0xAB, dtype=torch.uint8, device="cuda")
        ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr())
        kept[it] = out
    torch.cuda.synchronize()
    got = bytes(kept.cpu().numpy())
    for it in range(ITERS):
        assert got[64 * it:64 * it + 64] == sets[it % 3][2], it
    ctx.close()
