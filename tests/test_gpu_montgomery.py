"""GPU parity of SNARKV_FLAG_MONTGOMERY: every golden case re-encoded into halo2curves' in-memory form
(a * 2^256 mod r / mod p in 4 x u64 -- what `Fr`, `G1Affine`, `G2Affine` hold in memory; reference util/msm.rs:308 and
loader/native.rs:61-71 take those types) must give, after decoding, the bytes of the canonical call -- per-call flag and
context default, host-buffer, device-resident, batch, segmented, decide, deciding key, decompression, validation."""
import ctypes

import pytest

import bn254 as O
import coracle as C
import mont_util as M

pytestmark = pytest.mark.gpu


def _sv():
    import snark_verifier_amd as sv

    return sv


def test_golden_msms_in_memory_form(gpu_ctx, golden_msm):
    sv = _sv()
    F = sv.SNARKV_FLAG_MONTGOMERY
    for case in golden_msm:
        s, p, exp = bytes.fromhex(case["scalars"]), bytes.fromhex(case["points"]), bytes.fromhex(case["expected"])
        sm, pm = M.scalars_to_mont(s), M.coords_to_mont(p)
        for fn in (gpu_ctx.msm_naive, gpu_ctx.msm_pippenger):
            assert M.coords_from_mont(fn(sm, pm, F)) == exp, (case["name"], fn.__name__)
            assert fn(s, p) == exp  # the flag of one call leaves nothing behind
    # segmented: all cases in one launch
    s = b"".join(bytes.fromhex(c["scalars"]) for c in golden_msm)
    p = b"".join(bytes.fromhex(c["points"]) for c in golden_msm)
    offs = [0]
    for c in golden_msm:
        offs.append(offs[-1] + len(c["scalars"]) // 64)
    out = gpu_ctx.msm_batched(M.scalars_to_mont(s), M.coords_to_mont(p), offs, F)
    assert M.coords_from_mont(out) == b"".join(bytes.fromhex(c["expected"]) for c in golden_msm)


@pytest.mark.parametrize("chunks", ["1", "4", "16"])
@pytest.mark.parametrize("joint", ["0", "1"])
def test_every_term_kernel_in_memory_form(gpu_ctx, monkeypatch, chunks, joint):
    """the four term kernels of the segmented MSM (fixed window two-lane / grouped, one-lane and four-lane chains)"""
    sv = _sv()
    monkeypatch.setenv("SNARKV_NAIVE_CHUNKS", chunks)
    monkeypatch.setenv("SNARKV_NAIVE_JOINT", joint)
    n = 300
    s, p = C.sample_scalars(0xA1, n), C.sample_points(0xA2, n)
    offs = [0, 21, 24, 24 + 65, n]
    exp = C.msm_batched(s, p, offs)
    for quad in ("0", "1"):
        monkeypatch.setenv("SNARKV_NAIVE_QUAD", quad)
        got = gpu_ctx.msm_batched(M.scalars_to_mont(s), M.coords_to_mont(p), offs, sv.SNARKV_FLAG_MONTGOMERY)
        assert M.coords_from_mont(got) == exp, (chunks, joint, quad)


def test_context_default_covers_the_device_resident_entry_points():
    """snarkv_ctx_set_flags: the *_dev / *_many_* entry points carry no flags argument -- a context whose default is the
    in-memory form reads Montgomery inputs and writes Montgomery points there, incl. the chunk pipeline's lanes, the batch's
    job contexts, the partial + fold path and the device sampler (same elements, other encoding)."""
    import torch

    sv = _sv()
    ctx, ref = sv.Context(0), sv.Context(0)
    ctx.set_flags(sv.SNARKV_FLAG_MONTGOMERY)
    assert ctx.get_flags() == sv.SNARKV_FLAG_MONTGOMERY and ref.get_flags() == 0
    n = 70_000
    ds, dp = torch.empty(32 * n, dtype=torch.uint8, device="cuda"), torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    cs, cp = torch.empty_like(ds), torch.empty_like(dp)
    torch.cuda.synchronize()
    ctx.sample_scalars_dev(0x5EED0001, n, ds.data_ptr())
    ctx.sample_points_dev(0x5EED0002, n, dp.data_ptr())
    ref.sample_scalars_dev(0x5EED0001, n, cs.data_ptr())
    ref.sample_points_dev(0x5EED0002, n, cp.data_ptr())
    ctx.sync(), ref.sync()
    s, p = bytes(cs.cpu().numpy()), bytes(cp.cpu().numpy())
    assert s == C.sample_scalars(0x5EED0001, n) and p == C.sample_points(0x5EED0002, n)
    k = 2000  # the sampler's Montgomery output is the canonical stream re-encoded
    assert bytes(ds[:32 * k].cpu().numpy()) == M.scalars_to_mont(s[:32 * k])
    assert bytes(dp[:64 * k].cpu().numpy()) == M.coords_to_mont(p[:64 * k])
    exp = C.msm_pippenger(s, p, 8)
    out = torch.zeros(64 * 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr())
    ctx.sync()
    assert M.coords_from_mont(bytes(out[:64].cpu().numpy())) == exp
    # batch of three (job contexts), sizes that differ
    cuts = [(0, 30_000), (30_000, n), (100, 5_000)]
    ctx.msm_pippenger_many_dev([ds.data_ptr() + 32 * a for a, _ in cuts], [dp.data_ptr() + 64 * a for a, _ in cuts],
                               [b - a for a, b in cuts], out.data_ptr())
    ctx.sync()
    raw = bytes(out.cpu().numpy())
    for i, (a, b) in enumerate(cuts):
        assert M.coords_from_mont(raw[64 * i:64 * i + 64]) == C.msm_pippenger(s[32 * a:32 * b], p[64 * a:64 * b], 8), i
    # partial (opaque projective form: encoding-free) + fold (affine out: follows the flag)
    part = torch.zeros(sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    ctx.msm_pippenger_partial_dev(ds.data_ptr(), dp.data_ptr(), n, part.data_ptr())
    ctx.fold_partials_dev(part.data_ptr(), 1, out.data_ptr())
    ctx.sync()
    assert M.coords_from_mont(bytes(out[:64].cpu().numpy())) == exp
    ref.fold_partials_dev(part.data_ptr(), 1, out.data_ptr())
    ref.sync()
    assert bytes(out[:64].cpu().numpy()) == exp
    ctx.close(), ref.close()


def test_chunk_pipeline_in_memory_form(monkeypatch):
    """a large MSM (three chunks of the pipeline over shared bucket grids: its worker lanes inherit the encoding)"""
    import torch

    sv = _sv()
    monkeypatch.setenv("SNARKV_PIP_SPLIT", "2")
    ctx = sv.Context(0)
    n = (1 << 21) + 777
    s, p = C.sample_scalars(0xB1, n), C.sample_points(0xB2, n)
    exp = C.msm_pippenger(s, p, 16)
    assert ctx.msm_pippenger(s, p) == exp
    ctx.set_flags(sv.SNARKV_FLAG_MONTGOMERY)
    ds, dp = torch.empty(32 * n, dtype=torch.uint8, device="cuda"), torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sample_scalars_dev(0xB1, n, ds.data_ptr())
    ctx.sample_points_dev(0xB2, n, dp.data_ptr())
    ctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr())
    ctx.sync()
    assert M.coords_from_mont(bytes(out.cpu().numpy())) == exp
    ctx.close()


@pytest.mark.parametrize("form", ["1", "3"])
def test_decider_in_memory_form(gpu_ctx, golden_decider, form, monkeypatch):
    """deciding key (G2Affine coordinates) and accumulators in the in-memory form: same verdicts, same Gt bytes"""
    sv = _sv()
    monkeypatch.setenv("SNARKV_DECIDE_FORM", form)
    F = sv.SNARKV_FLAG_MONTGOMERY
    g1, g2, s_g2 = (bytes.fromhex(golden_decider[k]) for k in ("g1", "g2", "s_g2"))
    dkm = sv.DecidingKey(gpu_ctx, M.coords_to_mont(g1), M.coords_to_mont(g2), M.coords_to_mont(s_g2), F | sv.SNARKV_FLAG_VALIDATE)
    dkc = sv.DecidingKey(gpu_ctx, g1, g2, s_g2)
    cases = golden_decider["cases"]
    for case in cases:
        acc = bytes.fromhex(case["acc"])
        for dk in (dkm, dkc):  # the key is encoding-free once loaded
            assert gpu_ctx.decide(dk, M.coords_to_mont(acc), F) == case["accept"], case["name"]
            assert gpu_ctx.decide(dk, acc) == case["accept"], case["name"]
    accs = b"".join(bytes.fromhex(c["acc"]) for c in cases)
    allok, oks = gpu_ctx.decide_batch(dkm, M.coords_to_mont(accs), F | sv.SNARKV_FLAG_VALIDATE)
    assert oks == [c["accept"] for c in cases] and allok == all(oks)
    dkm.close(), dkc.close()


def test_validation_and_decompression_in_memory_form(gpu_ctx):
    sv = _sv()
    F = sv.SNARKV_FLAG_MONTGOMERY | sv.SNARKV_FLAG_VALIDATE
    n = 40
    s, p = C.sample_scalars(0xC1, n), C.sample_points(0xC2, n)
    exp = C.msm_pippenger(s, p, 2)
    sm, pm = M.scalars_to_mont(s), M.coords_to_mont(p)
    assert M.coords_from_mont(gpu_ctx.msm_pippenger(sm, pm, F)) == exp
    # an off-curve point / a non-reduced limb vector in the in-memory form is SNARKV_ERR_ENCODING
    bad = bytearray(pm)
    bad[64 * 7] ^= 1
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_pippenger(sm, bytes(bad), F)
    assert e.value.code == -3
    bad_s = bytearray(sm)
    bad_s[32 * 3:32 * 4] = (O.R + 5).to_bytes(32, "little")
    with pytest.raises(sv.SnarkvError) as e:
        gpu_ctx.msm_naive(bytes(bad_s), pm, F)
    assert e.value.code == -3
    # decompression: compressed input is the wire form, the points follow the context's flag
    comp = b""
    for i in range(n):
        x, y = p[64 * i:64 * i + 32], int.from_bytes(p[64 * i + 32:64 * i + 64], "little")
        comp += (int.from_bytes(x, "little") | ((y & 1) << 254)).to_bytes(32, "little")
    ctx = sv.Context(0)
    out, ok = ctx.g1_decompress(comp)
    assert all(ok) and out == p
    ctx.set_flags(sv.SNARKV_FLAG_MONTGOMERY)
    out, ok = ctx.g1_decompress(comp + (1 << 255).to_bytes(32, "little"))
    assert all(ok) and out == pm + bytes(64)
    # ... but a proof is wire data on both sides: snarkv_poseidon_read_batch answers in canonical bytes under either flag
    # (its points feed a transcript and the host's parser, include/snarkv_amd.h)
    import transcript as T

    spec = sv.PoseidonSpec(ctx, 5, 4, 8, 60, T.poseidon_opt_tables(5, 8, 60))
    recs = b"".join(comp[32 * i:32 * i + 32] + s[32 * i:32 * i + 32] for i in range(n))  # a record: one point, one scalar
    layout, seg = [(2 << 28) | 0, (3 << 28) | 0, (1 << 28) | 32], [3]
    ch_m, pts_m, ok_m = ctx.poseidon_read_batch(spec, recs, n, 64, b"", 0, layout, [0], seg)
    ctx.set_flags(0)
    ch_c, pts_c, ok_c = ctx.poseidon_read_batch(spec, recs, n, 64, b"", 0, layout, [0], seg)
    assert (ch_m, pts_m, ok_m) == (ch_c, pts_c, ok_c) and pts_c == p and all(ok_c)
    x0, y0 = int.from_bytes(p[:32], "little"), int.from_bytes(p[32:64], "little")
    assert int.from_bytes(ch_c[:32], "little") == T.poseidon_transcript_challenges([x0 % O.R, y0 % O.R, int.from_bytes(s[:32], "little")], seg)[0]
    spec.close()
    ctx.close()


def test_context_free_entry_points_in_memory_form(golden_msm, golden_decider):
    """bn254_set_flags: the process-global context speaks the in-memory form from then on (and back)"""
    sv = _sv()
    lib = sv.load_library()
    c = golden_msm[4]
    s, p, exp = bytes.fromhex(c["scalars"]), bytes.fromhex(c["points"]), bytes.fromhex(c["expected"])
    out = ctypes.create_string_buffer(64)
    try:
        assert lib.bn254_set_flags(sv.SNARKV_FLAG_MONTGOMERY) == 0
        assert lib.bn254_g1_msm_naive(M.scalars_to_mont(s), M.coords_to_mont(p), len(s) // 32, out) == 0
        assert M.coords_from_mont(out.raw) == exp
        assert lib.bn254_g1_msm_pippenger(M.scalars_to_mont(s), M.coords_to_mont(p), len(s) // 32, out) == 0
        assert M.coords_from_mont(out.raw) == exp
        g1, g2, s_g2 = (M.coords_to_mont(bytes.fromhex(golden_decider[k])) for k in ("g1", "g2", "s_g2"))
        for case in golden_decider["cases"][:4]:
            acc = M.coords_to_mont(bytes.fromhex(case["acc"]))
            assert lib.bn254_kzg_decide(g1, g2, s_g2, acc) == (1 if case["accept"] else 0), case["name"]
        assert lib.bn254_set_flags(4) == -5  # unknown bit
    finally:
        assert lib.bn254_set_flags(0) == 0
    assert lib.bn254_g1_msm_naive(s, p, len(s) // 32, out) == 0 and out.raw == exp
