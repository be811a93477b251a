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
