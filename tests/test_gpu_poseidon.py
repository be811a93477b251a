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
