"""GPU: batched Poseidon transcripts on the device (csrc/poseidon.hip through the C ABI
`snarkv_poseidon_create` / `snarkv_poseidon_transcript_batch`) vs the oracle sponge
(oracle/transcript.py; reference util/hash/poseidon.rs:115-202 and
system/halo2/transcript/halo2.rs:170-321)."""
import random

import pytest

import bn254 as O
import transcript as T

pytestmark = pytest.mark.gpu


def _spec(gpu_ctx, t, rate, r_f, r_p):
    import snark_verifier_amd as sv

    return sv.PoseidonSpec(gpu_ctx, t, rate, r_f, r_p, T.poseidon_opt_tables(t, r_f, r_p))


@pytest.mark.parametrize("params", [(5, 4, 8, 60), (3, 2, 8, 57), (2, 1, 8, 56), (8, 7, 8, 4), (4, 3, 6, 5)])
def test_transcript_batch_matches_the_oracle_sponge(gpu_ctx, params):
    t, rate, r_f, r_p = params
    spec = _spec(gpu_ctx, t, rate, r_f, r_p)
    rng = random.Random(sum(params))
    shapes = [[0], [1], [rate], [rate + 1], [0, 0], [2 * rate, 0, 1], [3, rate, 0, 2 * rate + 1, 5], [7, 0, 0, 9]]
    for seg in shapes:
        L = sum(seg)
        n = rng.choice([1, 2, 5, 33])
        rows = []
        for i in range(n):
            rows.append([rng.choice([0, 1, O.R - 1, rng.randrange(O.R)]) for _ in range(L)])
        elems = b"".join(x.to_bytes(32, "little") for row in rows for x in row)
        got = gpu_ctx.poseidon_transcript_batch(spec, elems, n, seg)
        for i, row in enumerate(rows):
            exp = T.poseidon_transcript_challenges(row, seg, t, rate, r_f, r_p)
            for q, e in enumerate(exp):
                o = 32 * (i * len(seg) + q)
                assert int.from_bytes(got[o:o + 32], "little") == e, (params, seg, i, q)
    spec.close()


def test_transcript_batch_of_a_real_proof_shape(gpu_ctx):
    """The element stream and squeeze positions of a StandardPlonk + GWC19 proof
    (instances, 6 witnesses in 3 phases, 3 quotient chunks, 19 evaluations, 4 openings),
    1 000 transcripts in one launch, spot-checked against the oracle."""
    spec = _spec(gpu_ctx, 5, 4, 8, 60)
    seg = [1 + 2 + 6, 0 + 0, 0, 6, 0, 6, 19, 8]  # initial state + instances + 3 points | theta | beta | gamma ... | z | v | u
    L, n = sum(seg), 1000
    rng = random.Random(5)
    rows = [[rng.randrange(O.R) for _ in range(L)] for _ in range(n)]
    elems = b"".join(x.to_bytes(32, "little") for row in rows for x in row)
    got = gpu_ctx.poseidon_transcript_batch(spec, elems, n, seg)
    for i in (0, 1, 499, 999):
        exp = T.poseidon_transcript_challenges(rows[i], seg)
        for q, e in enumerate(exp):
            o = 32 * (i * len(seg) + q)
            assert int.from_bytes(got[o:o + 32], "little") == e
    spec.close()


def test_error_codes(gpu_ctx):
    import snark_verifier_amd as sv

    spec = _spec(gpu_ctx, 5, 4, 8, 60)
    with pytest.raises(sv.SnarkvError):  # no transcripts
        gpu_ctx.poseidon_transcript_batch(spec, b"", 0, [0])
    with pytest.raises(sv.SnarkvError):  # t out of range
        sv.PoseidonSpec(gpu_ctx, 9, 8, 8, 60, T.poseidon_opt_tables(5, 8, 60))
    spec.close()


def test_g1_decompress_batch_vs_oracle(gpu_ctx):
    """`snarkv_g1_decompress` = `C::from_bytes` of PoseidonTranscript::read_ec_point (halo2.rs:260-273) for a batch:
    both parities, the identity, and every way an encoding can be invalid -- against the oracle's restatement."""
    import coracle as C
    import transcript as T

    n = 300
    pts = C.sample_points(71, n)
    enc, want, valid = [], [], []

    def push(b):
        b = bytes(b)
        enc.append(b)
        try:
            pt = T.g1_decompress(b)
            want.append(bytes(64) if pt is None else pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little"))
            valid.append(True)
        except T.TranscriptError:
            want.append(bytes(64))
            valid.append(False)

    for i in range(n):
        x = int.from_bytes(pts[64 * i:64 * i + 32], "little")
        y = int.from_bytes(pts[64 * i + 32:64 * i + 64], "little")
        push(T.g1_compress((x, y)))
        push(T.g1_compress((x, O.P - y)))  # the other root: the other parity
    assert want[0] == pts[:64]
    push(T.g1_compress(None))                                   # the identity
    bad_inf = bytearray(T.g1_compress(None)); bad_inf[0] = 1    # identity flag with a non-zero x
    push(bad_inf)
    bad_inf2 = bytearray(T.g1_compress(None)); bad_inf2[31] |= 0x40  # identity flag with the parity bit
    push(bad_inf2)
    big = bytearray(O.P.to_bytes(32, "little"))                 # x = p: not canonical
    push(big)
    big2 = bytearray((O.P + 5).to_bytes(32, "little"))
    push(big2)
    rng = random.Random(9)
    non_res = 0
    while non_res < 20:                                          # x with x^3 + 3 not a square: no such point
        x = rng.randrange(O.P)
        if pow((x * x * x + 3) % O.P, (O.P - 1) // 2, O.P) != 1:
            b = bytearray(x.to_bytes(32, "little"))
            b[31] |= rng.randrange(2) << 6
            push(b)
            non_res += 1
    for x in (0, 1, 2, O.P - 1):                                 # small / extreme x, whatever they decode to
        for s in (0, 1):
            b = bytearray(x.to_bytes(32, "little"))
            b[31] |= s << 6
            push(b)
    got, ok = gpu_ctx.g1_decompress(b"".join(enc))
    assert ok == valid
    for i, w in enumerate(want):
        assert got[64 * i:64 * i + 64] == w, i
    assert gpu_ctx.g1_decompress(b"") == (b"", [])


def test_read_batch_assembles_decompresses_and_hashes_like_the_pieces(gpu_ctx):
    """`snarkv_poseidon_read_batch`: the transcript input of proof-shaped byte records built ON THE DEVICE (lead elements,
    scalars at offsets, point coordinates mod r from the device's own decompression) == the oracle's sponge over the
    elements a native PoseidonTranscript absorbs (halo2.rs:215-275), incl. a coordinate >= r (one subtraction), an
    invalid point (flag 0; that record's challenges are meaningless and unchecked) and a lead-only segment."""
    import coracle as C

    spec = _spec(gpu_ctx, 5, 4, 8, 60)
    rng = random.Random(77)
    n, n_lead, stride = 37, 3, 7 * 32 + 16  # a record: scalar, point, point, scalar, scalar, point, scalar (+ 16 bytes of padding)
    kinds = "spPssps".replace("P", "p")
    pts_off = [32 * k for k, c in enumerate(kinds) if c == "p"]
    sc_off = [32 * k for k, c in enumerate(kinds) if c == "s"]
    # absorbed order: lead 0, lead 1 | point 0 (x, y), scalar 0 | lead 2, point 1, scalar 1, scalar 2, point 2, scalar 3 | (nothing)
    layout = [(0 << 28) | 0, (0 << 28) | 1, (2 << 28) | 0, (3 << 28) | 0, (1 << 28) | sc_off[0], (0 << 28) | 2, (2 << 28) | 1, (3 << 28) | 1,
              (1 << 28) | sc_off[1], (1 << 28) | sc_off[2], (2 << 28) | 2, (3 << 28) | 2, (1 << 28) | sc_off[3]]
    seg = [2, 3, 8, 0]
    raw = C.sample_points(0x99, 3 * n)
    # a point whose x is >= r (p > r: such x exist) so that the reduction is exercised
    big = None
    x = O.R
    while big is None:
        y2 = (x * x * x + 3) % O.P
        y = pow(y2, (O.P + 1) // 4, O.P)
        if y * y % O.P == y2:
            big = (x, y)
        x += 1
    recs, leads, exp_elems, exp_pts, exp_ok = [], [], [], [], []
    for i in range(n):
        P3 = []
        for q in range(3):
            o = 64 * (3 * i + q)
            P3.append((int.from_bytes(raw[o:o + 32], "little"), int.from_bytes(raw[o + 32:o + 64], "little")))
        if i == 5:
            P3[1] = big
        sc = [rng.choice([0, 1, O.R - 1, rng.randrange(O.R)]) for _ in range(4)]
        ld = [rng.randrange(O.R) for _ in range(n_lead)]
        enc = [T.g1_compress(p) for p in P3]
        okf = [1, 1, 1]
        if i == 9:  # an x with no point on the curve
            xb = 0
            while pow((xb ** 3 + 3) % O.P, (O.P - 1) // 2, O.P) == 1:
                xb += 1
            enc[2] = xb.to_bytes(32, "little")
            okf[2] = 0
        rec = bytearray(stride)
        it_p, it_s = iter(enc), iter(sc)
        for k, c in enumerate(kinds):
            rec[32 * k:32 * k + 32] = next(it_p) if c == "p" else next(it_s).to_bytes(32, "little")
        recs.append(bytes(rec))
        leads.append(b"".join(v.to_bytes(32, "little") for v in ld))
        exp_elems.append([ld[0], ld[1], P3[0][0] % O.R, P3[0][1] % O.R, sc[0], ld[2], P3[1][0] % O.R, P3[1][1] % O.R, sc[1], sc[2],
                          P3[2][0] % O.R, P3[2][1] % O.R, sc[3]])
        exp_pts.append(P3)
        exp_ok.append(okf)
    ch, pts, ok = gpu_ctx.poseidon_read_batch(spec, b"".join(recs), n, stride, b"".join(leads), n_lead, layout, pts_off, seg)
    assert list(ok) == [f for row in exp_ok for f in row]
    for i in range(n):
        for q in range(3):
            o = 64 * (3 * i + q)
            if exp_ok[i][q]:
                assert pts[o:o + 64] == exp_pts[i][q][0].to_bytes(32, "little") + exp_pts[i][q][1].to_bytes(32, "little"), (i, q)
            else:
                assert pts[o:o + 64] == bytes(64)
        if all(exp_ok[i]):
            exp = T.poseidon_transcript_challenges(exp_elems[i], seg)
            for q, e in enumerate(exp):
                o = 32 * (i * len(seg) + q)
                assert int.from_bytes(ch[o:o + 32], "little") == e, (i, q)
    # the argument checks
    import snark_verifier_amd as sv

    for bad in ([(1 << 28) | 2] + layout[1:], [(2 << 28) | 3] + layout[1:], [(0 << 28) | 3] + layout[1:]):
        with pytest.raises(sv.SnarkvError):  # a scalar offset not a multiple of 4, a point / lead index out of range
            gpu_ctx.poseidon_read_batch(spec, b"".join(recs), n, stride, b"".join(leads), n_lead, bad, pts_off, seg)
    with pytest.raises(sv.SnarkvError):  # a point offset not a multiple of 16
        gpu_ctx.poseidon_read_batch(spec, b"".join(recs), n, stride, b"".join(leads), n_lead, layout, [pts_off[0] + 4] + pts_off[1:], seg)
    # n * L beyond the kernels' 32-bit element index (n < 2^24 and L < 2^16 each pass): refused before anything is staged
    import array
    import ctypes

    big_n, big_l = 70000, 65000  # 4.55e9 elements
    lay, sg = array.array("I", [0] * big_l), array.array("I", [big_l])
    rc = gpu_ctx._lib.snarkv_poseidon_read_batch(gpu_ctx._h, spec._h, b"\x00" * 64, big_n, 64, b"\x00" * 32, 1,
                                                 ctypes.c_void_p(lay.buffer_info()[0]), big_l, None, 0,
                                                 ctypes.c_void_p(sg.buffer_info()[0]), 1, ctypes.create_string_buffer(32), None, None)
    assert rc == sv.SNARKV_ERR_LENGTH
    spec.close()
