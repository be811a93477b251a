"""GPU parity: HIP MSM kernels (through the C ABI) vs the oracle, bit-exact.

Covers `NativeLoader::multi_scalar_multiplication` (reference
snark-verifier/src/loader/native.rs:61-71), its segmented form, and
`util::msm::multi_scalar_multiplication` (reference util/msm.rs:308-343)."""
import pytest

import bn254 as O
import coracle as C

pytestmark = pytest.mark.gpu


def test_golden_naive(gpu_ctx, golden_msm):
    for case in golden_msm:
        s, p, exp = bytes.fromhex(case["scalars"]), bytes.fromhex(case["points"]), bytes.fromhex(case["expected"])
        assert gpu_ctx.msm_naive(s, p) == exp, case["name"]


def test_golden_pippenger(gpu_ctx, golden_msm):
    for case in golden_msm:
        s, p, exp = bytes.fromhex(case["scalars"]), bytes.fromhex(case["points"]), bytes.fromhex(case["expected"])
        assert gpu_ctx.msm_pippenger(s, p) == exp, case["name"]


def test_golden_batched_one_launch(gpu_ctx, golden_msm):
    s = b"".join(bytes.fromhex(c["scalars"]) for c in golden_msm)
    p = b"".join(bytes.fromhex(c["points"]) for c in golden_msm)
    offs = [0]
    for c in golden_msm:
        offs.append(offs[-1] + len(c["scalars"]) // 64)
    out = gpu_ctx.msm_batched(s, p, offs)
    for i, c in enumerate(golden_msm):
        assert out[64 * i:64 * i + 64] == bytes.fromhex(c["expected"]), c["name"]


def test_context_free_entry_points(golden_msm):
    import ctypes

    import snark_verifier_amd as sv

    lib = sv.load_library()
    c = golden_msm[3]
    s, p = bytes.fromhex(c["scalars"]), bytes.fromhex(c["points"])
    out = ctypes.create_string_buffer(64)
    assert lib.bn254_g1_msm_naive(s, p, len(s) // 32, out) == 0 and out.raw == bytes.fromhex(c["expected"])
    assert lib.bn254_g1_msm_pippenger(s, p, len(s) // 32, out) == 0 and out.raw == bytes.fromhex(c["expected"])


def test_error_codes(gpu_ctx, golden_msm):
    import snark_verifier_amd as sv

    # empty MSM: the reference panics (native.rs:69, msm.rs:265)
    for fn in (gpu_ctx.msm_naive, gpu_ctx.msm_pippenger):
        with pytest.raises(sv.SnarkvError) as e:
            fn(b"", b"")
        assert e.value.code == -1
    # length mismatch: assert_eq! at msm.rs:309
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_pippenger(b"\x00" * 64, b"\x00" * 64)
    assert e.value.code == -2
    # empty segment inside a batch
    c = golden_msm[1]
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_batched(bytes.fromhex(c["scalars"]), bytes.fromhex(c["points"]), [0, 2, 2])
    assert e.value.code == -1
    # validation: off-curve point, non-canonical scalar
    s, p = bytes.fromhex(c["scalars"]), bytearray(bytes.fromhex(c["points"]))
    p[40] ^= 1
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_naive(s, bytes(p), flags=sv.SNARKV_FLAG_VALIDATE)
    assert e.value.code == -3
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_naive(O.fe_to_bytes(O.R) + s[32:], bytes.fromhex(c["points"]), flags=sv.SNARKV_FLAG_VALIDATE)
    assert e.value.code == -3
    assert gpu_ctx.msm_naive(s, bytes.fromhex(c["points"]), flags=sv.SNARKV_FLAG_VALIDATE) == bytes.fromhex(c["expected"])


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 100, 1000, 4097])
def test_pippenger_vs_c_oracle_ragged(gpu_ctx, n):
    s, p = C.sample_scalars(100 + n, n), C.sample_points(200 + n, n)
    assert gpu_ctx.msm_pippenger(s, p) == C.msm_pippenger(s, p, 1)


def test_naive_vs_c_oracle_medium(gpu_ctx):
    n = 3000
    s, p = C.sample_scalars(5, n), C.sample_points(6, n)
    assert gpu_ctx.msm_naive(s, p) == C.msm_pippenger(s, p, 4)


def test_pippenger_2p16_vs_c_oracle(gpu_ctx):
    n = 1 << 16
    s, p = C.sample_scalars(0x5EED0001, n), C.sample_points(0x5EED0002, n)
    exp = C.msm_pippenger(s, p, 8)
    assert gpu_ctx.msm_pippenger(s, p) == exp


@pytest.mark.parametrize("n", [(1 << 17) + 3, 1 << 18, (1 << 19) - 5])
def test_pippenger_mid_sizes_vs_c_oracle(gpu_ctx, n):
    """the sizes where the default window size (13 -> 16) and the single-MSM run length (32 / 16 / 16 / 32 / 64 entries per
    lane at 2^16 .. 2^20 points) change: bytes ≡ C oracle, with and without the throughput hint (96-entry runs)"""
    import os

    import snark_verifier_amd as sv

    s, p = C.sample_scalars(0x5EED0001 + n, n), C.sample_points(0x5EED0002 + n, n)
    exp = C.msm_pippenger(s, p, os.cpu_count() or 1)
    assert gpu_ctx.msm_pippenger(s, p) == exp
    ctx = sv.Context(0)
    ctx.set_throughput_hint(True)
    assert ctx.msm_pippenger(s, p) == exp
    ctx.close()


def test_pippenger_skewed_scalars(gpu_ctx):
    """Non-uniform scalar distributions: the fixed-run accumulate must stay
    correct when single buckets span many runs."""
    n = 5000
    p = C.sample_points(77, n)
    for name, sc in (
        ("all_same", [0x1234567] * n),
        ("bits", [i & 1 for i in range(n)]),
        ("tiny", [i % 7 for i in range(n)]),
        ("r_minus_small", [O.R - 1 - (i % 3) for i in range(n)]),
    ):
        s = b"".join(O.fe_to_bytes(x) for x in sc)
        assert gpu_ctx.msm_pippenger(s, p) == C.msm_pippenger(s, p, 4), name


def test_pippenger_window_sizes(gpu_ctx):
    import torch

    n = 2000
    s, p = C.sample_scalars(31, n), C.sample_points(32, n)
    exp = C.msm_pippenger(s, p, 2)
    ds = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
    dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for c in (2, 3, 5, 8, 11, 13, 16):
        out.zero_()
        torch.cuda.synchronize()
        gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), window_bits=c)
        gpu_ctx.sync()
        assert bytes(out.cpu().numpy()) == exp, c


@pytest.mark.parametrize("quad", ["0", "1"])
def test_chunk_base_chain_one_lane_and_four_lanes(gpu_ctx, golden_msm, quad, monkeypatch):
    """the chunk-base doubling chain of the segmented kernel, one lane per (term, half) or a quad of lanes
    (jac29_double_quad): same bytes, every chunking, identity / zero-scalar / repeated-point terms included"""
    monkeypatch.setenv("SNARKV_NAIVE_QUAD", quad)
    n = 300
    s, p = bytearray(C.sample_scalars(51, n)), bytearray(C.sample_points(52, n))
    p[64 * 7:64 * 8] = bytes(64)            # an identity base
    s[32 * 9:32 * 10] = bytes(32)           # a zero scalar
    p[64 * 11:64 * 12] = p[64 * 10:64 * 11]  # a repeated base
    s, p = bytes(s), bytes(p)
    offs = [0, 1, 22, 25, 150, n]
    exp = C.msm_batched(s, p, offs)
    for chunks in ("2", "4", "8", "16"):
        monkeypatch.setenv("SNARKV_NAIVE_CHUNKS", chunks)
        assert gpu_ctx.msm_batched(s, p, offs) == exp, chunks
        assert gpu_ctx.msm_naive(s, p) == C.msm_pippenger(s, p, 2), chunks


@pytest.mark.parametrize("mode", ["0", "1", "2", "3", "4"])
def test_fixed_window_term_kernels_two_lane_and_group(gpu_ctx, golden_msm, mode, monkeypatch):
    """The one-launch term kernels of the segmented MSM (SNARKV_NAIVE_CHUNKS = 1): two lanes per term (0), and SEVERAL TERMS OF
    A SEGMENT per lane on shared doublings with affine tables (group: 1 = the device picks K, 2 / 3 / 4 force K).  Same
    bytes as the oracle on segments shorter, equal to and longer than K, with identity bases, zero
    scalars, a base listed twice with the same and with the opposite scalar inside one group (the degenerate fall-back),
    scalars built from the GLV lambda, and on every golden case in one launch."""
    monkeypatch.setenv("SNARKV_NAIVE_CHUNKS", "1")
    monkeypatch.setenv("SNARKV_NAIVE_JOINT", mode)
    n = 700
    s, p = bytearray(C.sample_scalars(61, n)), bytearray(C.sample_points(62, n))
    p[64 * 7:64 * 8] = bytes(64)              # an identity base
    s[32 * 9:32 * 10] = bytes(32)             # a zero scalar
    p[64 * 11:64 * 12] = p[64 * 10:64 * 11]   # a repeated base ...
    s[32 * 11:32 * 12] = s[32 * 10:32 * 11]   # ... with the same scalar: P + P inside a group
    p[64 * 13:64 * 14] = p[64 * 12:64 * 13]
    s[32 * 13:32 * 14] = ((O.R - int.from_bytes(s[32 * 12:32 * 13], "little")) % O.R).to_bytes(32, "little")  # ... and -k: cancels
    lam = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23  # a cube root of unity mod r
    assert pow(lam, 3, O.R) == 1
    for i, k in enumerate((lam, O.R - lam, lam + 1, (lam * lam) % O.R, 1, O.R - 1, 4, 8)):
        s[32 * (20 + i):32 * (21 + i)] = k.to_bytes(32, "little")
    s, p = bytes(s), bytes(p)
    offs = [0, 1, 3, 6, 10, 15, 15 + 21, 15 + 24, 100, 101, 300, n]
    assert gpu_ctx.msm_batched(s, p, offs) == C.msm_batched(s, p, offs)
    assert gpu_ctx.msm_naive(s, p) == C.msm_pippenger(s, p, 2)
    s = b"".join(bytes.fromhex(c["scalars"]) for c in golden_msm)
    p = b"".join(bytes.fromhex(c["points"]) for c in golden_msm)
    offs = [0]
    for c in golden_msm:
        offs.append(offs[-1] + len(c["scalars"]) // 64)
    assert gpu_ctx.msm_batched(s, p, offs) == b"".join(bytes.fromhex(c["expected"]) for c in golden_msm)


def test_group_kernel_walks_more_lanes_than_the_grid(gpu_ctx, monkeypatch):
    """more segments than one block of the lane map (150 000 terms in segments of 1 .. 9, against the oracle), and more lane
    groups than resident wavefronts -- the blocks walk their share -- (400 000 terms in pairs with K = 2: 200 000 lanes on a
    grid of at most 131 072; against the two-lane kernel, itself pinned to the oracle above)"""
    import torch

    monkeypatch.setenv("SNARKV_NAIVE_CHUNKS", "1")
    monkeypatch.setenv("SNARKV_NAIVE_JOINT", "1")
    n = 150_000
    s, p = C.sample_scalars(71, n), C.sample_points(72, n)
    offs, k = [0], 0
    while offs[-1] < n:
        offs.append(min(n, offs[-1] + 1 + (k * 7) % 9))
        k += 1
    assert gpu_ctx.msm_batched(s, p, offs) == C.msm_batched(s, p, offs)
    n = 400_000
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    do = torch.arange(0, n + 1, 2, dtype=torch.int32, device="cuda")
    out = [torch.zeros(64 * (n // 2), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    gpu_ctx.sample_scalars_dev(73, n, ds.data_ptr())
    gpu_ctx.sample_points_dev(74, n, dp.data_ptr())
    for mode, o in (("2", out[0]), ("0", out[1])):
        monkeypatch.setenv("SNARKV_NAIVE_JOINT", mode)
        gpu_ctx.msm_batched_dev(ds.data_ptr(), dp.data_ptr(), do.data_ptr(), n // 2, n, o.data_ptr())
        gpu_ctx.sync()
    assert torch.equal(out[0], out[1]) and int(out[0].count_nonzero()) > 60 * (n // 2)


def test_device_sampler_matches_oracle(gpu_ctx):
    import torch

    n = 777
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.sample_scalars_dev(0xABC, n, ds.data_ptr(), first=13)
    gpu_ctx.sample_points_dev(0xDEF, n, dp.data_ptr(), first=13)
    gpu_ctx.sync()
    assert bytes(ds.cpu().numpy()) == C.sample_scalars(0xABC, n, first=13)
    assert bytes(dp.cpu().numpy()) == C.sample_points(0xDEF, n, first=13)


def test_full_size_2p20_properties(gpu_ctx):
    """BASELINE config 2 size (2^20), checked through size-independent
    properties: (1) linearity -- MSM over the whole = fold of the MSMs over two
    uneven shards (the multi-GPU combine path); (2) determinism; (3) a
    2^12-point prefix agrees with the C oracle."""
    import torch

    import snark_verifier_amd as sv

    n = 1 << 20
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.sample_scalars_dev(0x5EED0001, n, ds.data_ptr())
    gpu_ctx.sample_points_dev(0x5EED0002, n, dp.data_ptr())
    out = torch.zeros(2, 64, dtype=torch.uint8, device="cuda")
    parts = torch.zeros(2, sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the context runs on its own stream
    gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out[0].data_ptr())
    gpu_ctx.sync()
    full = bytes(out[0].cpu().numpy())
    gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out[1].data_ptr())
    gpu_ctx.sync()
    assert bytes(out[1].cpu().numpy()) == full and full != b"\x00" * 64
    assert C.g1_is_on_curve(full)
    cut = 333_333
    gpu_ctx.msm_pippenger_partial_dev(ds.data_ptr(), dp.data_ptr(), cut, parts[0].data_ptr())
    gpu_ctx.msm_pippenger_partial_dev(ds.data_ptr() + 32 * cut, dp.data_ptr() + 64 * cut, n - cut, parts[1].data_ptr())
    gpu_ctx.fold_partials_dev(parts.data_ptr(), 2, out[1].data_ptr())
    gpu_ctx.sync()
    assert bytes(out[1].cpu().numpy()) == full
    m = 1 << 12
    s = bytes(ds[: 32 * m].cpu().numpy())
    p = bytes(dp[: 64 * m].cpu().numpy())
    gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), m, out[1].data_ptr())
    gpu_ctx.sync()
    assert bytes(out[1].cpu().numpy()) == C.msm_pippenger(s, p, 4)


@pytest.mark.parametrize("chunks", [1, 2, 4, 8, 16])
def test_batched_chunked_scalar_mul_all_chunkings(gpu_ctx, golden_msm, chunks, monkeypatch):
    """The small-batch form of the naive MSM (msm_naive.hip K1a/K1b: one doubling
    chain + J chunk lanes per half-scalar) must agree with the oracle for every
    chunk count, on the goldens and on edge scalars / identity bases."""
    import random

    monkeypatch.setenv("SNARKV_NAIVE_CHUNKS", str(chunks))
    s = b"".join(bytes.fromhex(c["scalars"]) for c in golden_msm)
    p = b"".join(bytes.fromhex(c["points"]) for c in golden_msm)
    offs = [0]
    for c in golden_msm:
        offs.append(offs[-1] + len(c["scalars"]) // 64)
    out = gpu_ctx.msm_batched(s, p, offs)
    for i, c in enumerate(golden_msm):
        assert out[64 * i:64 * i + 64] == bytes.fromhex(c["expected"]), (chunks, c["name"])

    rng = random.Random(100 + chunks)
    edge = [0, 1, 2, 3, O.R - 1, O.R - 2, (O.R - 1) // 2, 1 << 127, (1 << 127) - 1, 1 << 128, (1 << 253) + 5,
            0xFFFF, 0x10000, 0xFF00FF00FF00FF00FF00FF00FF00FF00, 1 << 16, 1 << 32, 1 << 64, 1 << 96]
    scal = edge + [rng.randrange(O.R) for _ in range(46)]
    pts = C.sample_points(900 + chunks, len(scal))
    pl = [pts[64 * i:64 * i + 64] for i in range(len(scal))]
    pl[5] = bytes(64)                       # identity base
    pl[20] = pl[19]                         # repeated base
    pl[22] = O.g1_to_bytes(O.g1_neg(O.g1_from_bytes(pl[21])))  # opposite bases
    scal[22] = scal[21]                     # ... with equal scalars: the pair cancels
    sb = b"".join(x.to_bytes(32, "little") for x in scal)
    pb = b"".join(pl)
    offs2 = [0, 1, 2, 18, 19, 23, 40, len(scal)]
    got = gpu_ctx.msm_batched(sb, pb, offs2)
    assert got == C.msm_batched(sb, pb, offs2)
    # single-term segments: every edge scalar on its own
    offs3 = list(range(len(scal) + 1))
    assert gpu_ctx.msm_batched(sb, pb, offs3) == C.msm_batched(sb, pb, offs3)


def test_large_msm_chunk_pipelined_form_matches_unsplit(gpu_ctx, monkeypatch):
    """n > 2^21 runs as 2^20-point chunks on worker lanes that add their bucket sums into shared grids, with ONE
    bucket reduce / shift / to_affine at the end (capi.hip pippenger_maybe_split): same bytes as the single-launch
    form, for an exact multiple and a ragged size, affine result and projective partial."""
    import torch

    import snark_verifier_amd as sv

    n = (1 << 22) + 12345
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    gpu_ctx.sample_scalars_dev(31, n, ds.data_ptr())
    gpu_ctx.sample_points_dev(32, n, dp.data_ptr())
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    res = {}
    for split in ("0", "1"):
        monkeypatch.setenv("SNARKV_PIP_SPLIT", split)
        for m in (1 << 22, n):
            gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), m, out.data_ptr(), 0)
            gpu_ctx.sync()
            res[(split, m)] = bytes(out.cpu().numpy())
    for m in (1 << 22, n):
        assert res[("0", m)] == res[("1", m)] != bytes(64)
    # two chunks (the second one ragged) through the shared bucket grid: opt-in below the default threshold of three
    monkeypatch.setenv("SNARKV_PIP_SPLIT", "2")
    m2 = (1 << 20) + 4099
    gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), m2, out.data_ptr(), 0)
    gpu_ctx.sync()
    two = bytes(out.cpu().numpy())
    monkeypatch.setenv("SNARKV_PIP_SPLIT", "0")
    gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), m2, out.data_ptr(), 0)
    gpu_ctx.sync()
    assert two == bytes(out.cpu().numpy()) != bytes(64)
    # partial + fold (the multi-GPU building blocks) through the split path
    monkeypatch.setenv("SNARKV_PIP_SPLIT", "1")
    part = torch.zeros(sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    gpu_ctx.msm_pippenger_partial_dev(ds.data_ptr(), dp.data_ptr(), n, part.data_ptr(), 0)
    gpu_ctx.fold_partials_dev(part.data_ptr(), 1, out.data_ptr())
    gpu_ctx.sync()
    assert bytes(out.cpu().numpy()) == res[("1", n)]


def test_largest_config_2p24_split_linearity(gpu_ctx):
    """BASELINE's largest size (2^24 points), the size-independent property: MSM(all) == MSM(first part) + MSM(rest),
    through the projective-partial + fold entry points the multi-GPU path uses, at an uneven split (both parts take the
    chunk pipeline).  The bit-for-bit comparison with the threaded C oracle at this size is
    tests/test_gpu_fullsize.py::test_config4_2p24_bit_exact_vs_c_oracle."""
    import torch

    import snark_verifier_amd as sv

    n = 1 << 24
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    gpu_ctx.sample_scalars_dev(0x5EED0001, n, ds.data_ptr())
    gpu_ctx.sample_points_dev(0x5EED0002, n, dp.data_ptr())
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    gpu_ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr(), 0)
    h = (n // 3) | 1
    parts = torch.zeros(2 * sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    gpu_ctx.msm_pippenger_partial_dev(ds.data_ptr(), dp.data_ptr(), h, parts.data_ptr(), 0)
    gpu_ctx.msm_pippenger_partial_dev(ds.data_ptr() + 32 * h, dp.data_ptr() + 64 * h, n - h,
                                      parts.data_ptr() + sv.G1_PARTIAL_BYTES, 0)
    out2 = torch.zeros(64, dtype=torch.uint8, device="cuda")
    gpu_ctx.fold_partials_dev(parts.data_ptr(), 2, out2.data_ptr())
    gpu_ctx.sync()
    a = bytes(out.cpu().numpy())
    assert a == bytes(out2.cpu().numpy()) and a != bytes(64)
    del ds, dp
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,world", [(1, 2), (777, 2), (5000, 3), (1 << 16, 4), (4097, 11)])
def test_bucket_sharded_building_blocks_emulated_ranks(gpu_ctx, n, world):
    """The "bucket-sum allreduce" variant emulated on one device: `world` point
    shards fill the GLOBAL bucket grid, the grids are added, each emulated rank reduces the
    windows it owns, the partials are folded -- equal to the C oracle's Pippenger.  Identical
    points appear in several shards (cut into a repeated pattern) so the careful bucket add
    meets doublings; world=11 > windows/2 exercises uneven and empty window ranges."""
    import torch

    import snark_verifier_amd as sv
    from snark_verifier_amd.distributed import shard_range

    s = C.sample_scalars(0x51, n)
    base = C.sample_points(0x52, max(1, n // 3))
    p = (base * 4)[: 64 * n]  # every point repeated across the shards
    c, W, B = sv.Context.bucket_geometry(n)
    assert W == -(-128 // c) and B == 1 << (c - 1)
    ds = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
    dp = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
    PB = sv.G1_PARTIAL_BYTES
    grids = torch.zeros(world, W * B * PB, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        if hi > lo:
            gpu_ctx.fill_buckets_dev(ds.data_ptr() + 32 * lo, dp.data_ptr() + 64 * lo, hi - lo, c, grids[r].data_ptr())
    for r in range(1, world):
        gpu_ctx.buckets_add_dev(grids[0].data_ptr(), grids[r].data_ptr(), W * B)
    parts = torch.zeros(world, PB, dtype=torch.uint8, device="cuda")
    gpu_ctx.sync()
    torch.cuda.synchronize()
    for r in range(world):
        w0, w1 = shard_range(W, r, world)
        if w1 > w0:
            gpu_ctx.buckets_reduce_dev(grids[0].data_ptr() + w0 * B * PB, c, w0, w1 - w0, parts[r].data_ptr())
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    gpu_ctx.fold_partials_dev(parts.data_ptr(), world, out.data_ptr())
    gpu_ctx.sync()
    assert bytes(out.cpu().numpy()) == C.msm_pippenger(s, p, 4)


def test_bucket_sharded_error_paths(gpu_ctx):
    import torch

    import snark_verifier_amd as sv

    d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(sv.SnarkvError):
        gpu_ctx.fill_buckets_dev(d.data_ptr(), d.data_ptr(), 0, 8, d.data_ptr())  # empty
    with pytest.raises(sv.SnarkvError):
        gpu_ctx.fill_buckets_dev(d.data_ptr(), d.data_ptr(), 4, 0, d.data_ptr())  # c must be explicit
    with pytest.raises(sv.SnarkvError):
        gpu_ctx.buckets_reduce_dev(d.data_ptr(), 8, 15, 2, d.data_ptr())  # window range past the top window
    with pytest.raises(sv.SnarkvError):
        gpu_ctx.buckets_add_dev(d.data_ptr(), d.data_ptr(), 0)


def test_throughput_hint_gives_the_same_bytes():
    """`snarkv_ctx_set_throughput_hint` changes how the sorted stream is cut into per-lane runs (96 instead of 64 entries)
    -- never the result: sizes around the run / tile boundaries, skewed scalars (one bucket spanning many runs), and the
    2^20 workload against the C oracle."""
    import os

    import torch

    import snark_verifier_amd as sv

    ctx = sv.Context(0)
    ctx.set_throughput_hint(True)
    for n in (1, 95, 96, 97, 4097, 50_000):
        s, p = C.sample_scalars(0x700 + n, n), C.sample_points(0x701 + n, n)
        assert ctx.msm_pippenger(s, p) == C.msm_pippenger(s, p, 8), n
    n = 5000
    p = C.sample_points(77, n)
    s = b"".join(O.fe_to_bytes(x) for x in [0x1234567] * n)
    assert ctx.msm_pippenger(s, p) == C.msm_pippenger(s, p, 4)
    n = 1 << 20
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sample_scalars_dev(0x5EED0001, n, ds.data_ptr())
    ctx.sample_points_dev(0x5EED0002, n, dp.data_ptr())
    ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr())
    ctx.sync()
    hinted = bytes(out.cpu().numpy())
    ctx.set_throughput_hint(False)
    ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr())
    ctx.sync()
    assert hinted == bytes(out.cpu().numpy())
    assert hinted == C.msm_pippenger(bytes(ds.cpu().numpy()), bytes(dp.cpu().numpy()), os.cpu_count() or 1)
    ctx.close()
