"""GPU: fixed-base rows of the segmented MSM (csrc/msm_fixed.hip; include/snarkv_amd.h `snarkv_g1_fixed_table_*`,
`snarkv_g1_msm_batched_fixed`): segments whose terms on the protocol's shared bases (preprocessed commitments + generator:
9 of the 21 terms of Gwc19's left MSM, gwc19.rs:124-139) are evaluated from window tables -- byte for byte the plain
segmented MSM (`NativeLoader::multi_scalar_multiplication`, native.rs:61-71) of the C oracle on the same (scalar, point)
pairs: every kernel regime of the variable part, both encodings, identity / duplicate / opposite fixed bases, edge scalars,
segments with only fixed or only variable terms."""
import random

import pytest

import bn254 as O
import coracle as C
import mont_util as M

pytestmark = pytest.mark.gpu


def _mixed_job(rng, tab_points, nseg, nvar, nfix, seed):
    """nseg segments of nvar variable + nfix fixed terms (fixed ids drawn from the table); returns the arguments of the
    fixed call and the equivalent plain (scalars, points, offsets)"""
    nb = len(tab_points) // 64
    vs, vp = C.sample_scalars(seed, nseg * nvar), C.sample_points(seed + 1, nseg * nvar)
    fs = C.sample_scalars(seed + 2, nseg * nfix)
    ids = [rng.randrange(nb) for _ in range(nseg * nfix)]
    offs = [k * nvar for k in range(nseg + 1)]
    foffs = [k * nfix for k in range(nseg + 1)]
    ps, pp, po = bytearray(), bytearray(), [0]
    for k in range(nseg):
        ps += vs[32 * k * nvar:32 * (k + 1) * nvar]
        pp += vp[64 * k * nvar:64 * (k + 1) * nvar]
        for i in range(k * nfix, (k + 1) * nfix):
            ps += fs[32 * i:32 * i + 32]
            pp += tab_points[64 * ids[i]:64 * ids[i] + 64]
        po.append(po[-1] + nvar + nfix)
    return (vs, vp, offs, fs, ids, foffs), (bytes(ps), bytes(pp), po)


@pytest.fixture(scope="module")
def table9(gpu_ctx):
    import snark_verifier_amd as sv

    pts = C.sample_points(0xF1ED, 8) + O.g1_to_bytes(O.G1_GEN)  # 8 "preprocessed commitments" + the generator
    tab = sv.FixedTable(gpu_ctx, pts)
    assert tab.n == 9
    yield pts, tab
    tab.close()


@pytest.mark.parametrize("nseg,nvar,nfix", [(1, 12, 9), (64, 12, 9), (70, 3, 0), (33, 0, 9), (700, 12, 9), (1500, 15, 9), (2100, 12, 9)])
def test_gwc19_shapes_every_regime_of_the_variable_part(gpu_ctx, table9, nseg, nvar, nfix):
    """64 / 700 / 1500 / 2100 proofs' left MSMs: the variable part runs chunked x16, x4, one lane pair per term, ..."""
    pts, tab = table9
    rng = random.Random(nseg * 100 + nvar)
    fixed_args, plain = _mixed_job(rng, pts, nseg, nvar, nfix, 0x6000 + nseg)
    got = gpu_ctx.msm_batched_fixed(tab, *fixed_args)
    assert got == C.msm_batched(*plain)


def test_throughput_regime_grouped_kernel(gpu_ctx, table9):
    """>= 49 152 variable terms: the grouped (Straus) kernel next to the fixed rows"""
    pts, tab = table9
    rng = random.Random(7)
    fixed_args, plain = _mixed_job(rng, pts, 4200, 12, 9, 0x6A00)
    assert gpu_ctx.msm_batched_fixed(tab, *fixed_args) == C.msm_batched(*plain)


def test_edge_scalars_and_exceptional_bases(gpu_ctx):
    """table bases: B, B again, -B, the identity, the generator; scalars 0, 1, r - 1, 128 * 256^j (digit boundaries),
    0x80..80 / 0x81..81 (carry chains), 2^253; B and -B with the same scalar in one segment cancel"""
    import snark_verifier_amd as sv

    b = C.sample_points(0xB0, 1)
    neg = O.g1_to_bytes(O.g1_neg(O.g1_from_bytes(b)))
    pts = b + b + neg + bytes(64) + O.g1_to_bytes(O.G1_GEN)
    tab = sv.FixedTable(gpu_ctx, pts)
    R = O.R
    edge = [0, 1, 2, 127, 128, 129, 255, 256, R - 1, R - 2, 1 << 253, (1 << 253) + 128, int.from_bytes(b"\x80" * 31, "little"),
            int.from_bytes(b"\x81" * 31, "little"), int.from_bytes(b"\x7f" * 31 + b"\x2f", "little"), 128 << 240, 129 << 240,
            (1 << 248) * 47, 0x80 << 8, 0xFF << 8, R // 2, R // 3]
    fs = b"".join(O.fe_to_bytes(x) for x in edge)
    for bid in range(5):
        n = len(edge)
        ids = [bid] * n
        # one fixed term per segment, plus one variable term so that both paths run
        vs, vp = C.sample_scalars(0xE0 + bid, n), C.sample_points(0xE8 + bid, n)
        offs, foffs = list(range(n + 1)), list(range(n + 1))
        got = gpu_ctx.msm_batched_fixed(tab, vs, vp, offs, fs, ids, foffs)
        ps = b"".join(vs[32 * i:32 * i + 32] + fs[32 * i:32 * i + 32] for i in range(n))
        pp = b"".join(vp[64 * i:64 * i + 64] + pts[64 * bid:64 * bid + 64] for i in range(n))
        assert got == C.msm_batched(ps, pp, [2 * i for i in range(n + 1)]), bid
    # s B + s (-B) + s' B(again) in ONE segment, no variable term: = s' B
    s, s2 = O.fe_to_bytes(0x1234567890ABCDEF << 100), O.fe_to_bytes(R - 5)
    got = gpu_ctx.msm_batched_fixed(tab, b"", b"", [0, 0], s + s + s2, [0, 2, 1], [0, 3])
    assert got == C.msm_batched(s2, b, [0, 1])
    # everything cancels: the identity (64 zero bytes)
    got = gpu_ctx.msm_batched_fixed(tab, b"", b"", [0, 0], s + s, [1, 2], [0, 2])
    assert got == bytes(64)
    tab.close()


def test_in_memory_form_table_terms_and_results(gpu_ctx, table9):
    """SNARKV_FLAG_MONTGOMERY: bases at table creation, scalars and variable points of the call, results -- all as
    halo2curves holds them; a table built in one encoding serves calls in the other"""
    import snark_verifier_amd as sv

    pts, tab = table9
    rng = random.Random(11)
    (vs, vp, offs, fs, ids, foffs), plain = _mixed_job(rng, pts, 40, 12, 9, 0x6B00)
    exp = C.msm_batched(*plain)
    flag = sv.SNARKV_FLAG_MONTGOMERY
    got = gpu_ctx.msm_batched_fixed(tab, M.scalars_to_mont(vs), M.coords_to_mont(vp), offs, M.scalars_to_mont(fs), ids, foffs, flags=flag)
    assert got == M.coords_to_mont(exp)
    mtab = sv.FixedTable(gpu_ctx, M.coords_to_mont(pts), flags=flag)
    assert gpu_ctx.msm_batched_fixed(mtab, vs, vp, offs, fs, ids, foffs) == exp
    assert gpu_ctx.msm_batched_fixed(mtab, M.scalars_to_mont(vs), M.coords_to_mont(vp), offs, M.scalars_to_mont(fs), ids, foffs,
                                     flags=flag) == M.coords_to_mont(exp)
    mtab.close()


def test_errors_and_the_context_free_form(gpu_ctx, table9):
    import ctypes

    import snark_verifier_amd as sv

    pts, tab = table9
    s1, p1 = C.sample_scalars(1, 2), C.sample_points(2, 2)
    with pytest.raises(sv.SnarkvError) as e:  # a segment with neither kind of term: the reference panics (native.rs:69)
        gpu_ctx.msm_batched_fixed(tab, s1, p1, [0, 2, 2], s1[:32], [0], [0, 1, 1])
    assert e.value.code == sv.SNARKV_ERR_EMPTY
    with pytest.raises(sv.SnarkvError) as e:  # base id outside the table
        gpu_ctx.msm_batched_fixed(tab, s1, p1, [0, 2], s1[:32], [9], [0, 1])
    assert e.value.code == sv.SNARKV_ERR_ARG and "base 9" in str(e.value)
    with pytest.raises(sv.SnarkvError) as e:
        sv.FixedTable(gpu_ctx, b"")
    assert e.value.code == sv.SNARKV_ERR_EMPTY
    bad = bytearray(pts)
    bad[3] ^= 1
    with pytest.raises(sv.SnarkvError) as e:  # an off-curve base with SNARKV_FLAG_VALIDATE
        sv.FixedTable(gpu_ctx, bytes(bad), flags=sv.SNARKV_FLAG_VALIDATE)
    assert e.value.code == sv.SNARKV_ERR_ENCODING
    # context-free (trait-boundary) form
    lib = sv.load_library()
    h = ctypes.c_void_p()
    assert lib.bn254_g1_fixed_table_create(pts, 9, ctypes.byref(h)) == 0
    rng = random.Random(5)
    (vs, vp, offs, fs, ids, foffs), plain = _mixed_job(rng, pts, 20, 12, 9, 0x6C00)
    out = ctypes.create_string_buffer(64 * 20)
    o, fo, fi = (ctypes.c_uint32 * 21)(*offs), (ctypes.c_uint32 * 21)(*foffs), (ctypes.c_uint32 * len(ids))(*ids)
    assert lib.bn254_g1_msm_batched_fixed(h, vs, vp, o, fs, fi, fo, 20, out) == 0
    assert out.raw == C.msm_batched(*plain)
    lib.snarkv_g1_fixed_table_destroy(h)
