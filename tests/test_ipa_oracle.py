"""CPU: the IPA oracle (oracle/ipa.py, restating pcs/ipa.rs + pcs/ipa/{accumulation,decider}.rs)
against the committed fixture and its own algebra -- the reference's `test_ipa` / `test_ipa_as`
flows (prove -> read_proof -> succinct_verify -> decide; 10 accumulators -> IpaAs -> decide) with a
seeded RNG instead of OsRng, plus the rejections the reference does not test."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bn254 as O  # noqa: E402
import ipa as I  # noqa: E402
import transcript as T  # noqa: E402
from ipa_util import acc_from_json, bgh19_case, case_key, load_cases  # noqa: E402

TR = {"evm": T.EvmTranscript, "poseidon": T.PoseidonTranscript}


def test_h_eval_is_the_evaluation_of_h_coeffs():
    rnd = random.Random(1)
    for k in (1, 2, 5, 9):
        xi = [rnd.randrange(O.R) for _ in range(k)]
        z = rnd.randrange(O.R)
        h = I.h_coeffs(xi, 1)
        assert len(h) == 1 << k and h[0] == 1
        assert I.h_eval(xi, z) == I.poly_eval(h, z)
        s = rnd.randrange(O.R)
        assert I.h_coeffs(xi, s) == [c * s % O.R for c in h]
    with pytest.raises(AssertionError):
        I.h_coeffs([], 1)  # ipa.rs:406


@pytest.mark.parametrize("idx", range(4))
def test_golden_openings_and_accumulation(idx):
    c = load_cases()[idx]
    g, h, s = case_key(c)
    Tr, zk, k = TR[c["transcript"]], c["zk"], c["k"]
    accs = []
    for o in c["openings"]:
        com = O.g1_from_bytes(bytes.fromhex(o["commitment"]))
        z, ev = (O.fe_from_bytes(bytes.fromhex(o[n])) for n in ("z", "eval"))
        proof = bytes.fromhex(o["proof"])
        acc = I.ipa_succinct_verify(h, s, [(1, com)], z, ev, I.ipa_read_proof(zk, k, Tr(proof)))
        exp = acc_from_json(o["accumulator"])
        assert (acc[0], acc[1]) == (exp[0], exp[1])
        assert I.ipa_decide(g, acc)
        accs.append(acc)
        # a wrong evaluation fails the succinct check; a wrong U passes it only to fail `decide`
        with pytest.raises(I.IpaError):
            I.ipa_succinct_verify(h, s, [(1, com)], z, (ev + 1) % O.R, I.ipa_read_proof(zk, k, Tr(proof)))
        assert not I.ipa_decide(g, (acc[0], O.g1_add(acc[1], h)))
        assert not I.ipa_decide(g, ([(acc[0][0] + 1) % O.R] + acc[0][1:], acc[1]))
        # truncated proof -> Transcript error
        with pytest.raises(T.TranscriptError):
            I.ipa_read_proof(zk, k, Tr(proof[:-1]))
    as_proof = bytes.fromhex(c["as_proof"])
    acc = I.ipa_as_verify(h, s, accs, I.ipa_as_read_proof(zk, k, accs, Tr(as_proof)))
    exp = acc_from_json(c["as_accumulator"])
    assert (acc[0], acc[1]) == (exp[0], exp[1]) and I.ipa_decide(g, acc)
    # accumulating a BAD old accumulator still passes the succinct check (that is the point of
    # deferring) but the new accumulator differs and the honest proof no longer verifies
    bad = [accs[0], (accs[1][0], O.g1_add(accs[1][1], h)), accs[2]]
    with pytest.raises(I.IpaError):
        I.ipa_as_verify(h, s, bad, I.ipa_as_read_proof(zk, k, bad, Tr(as_proof)))
    with pytest.raises(AssertionError):
        I.ipa_as_read_proof(zk, k, accs[:1], Tr(as_proof))  # accumulation.rs:107: needs > 1 instances


def test_accumulation_of_ten_like_the_reference_test():
    """accumulation.rs:240-290 (`test_ipa_as`: zk, 10 accumulators) at k = 4."""
    import coracle as C

    rnd = random.Random(77)
    rng = lambda: rnd.randrange(O.R)  # noqa: E731
    k = 4
    raw = C.sample_points(123, (1 << k) + 2)
    pts = [O.g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range((1 << k) + 2)]
    pk = I.IpaProvingKey(k, pts[:1 << k], pts[1 << k], pts[(1 << k) + 1])
    accs = []
    for _ in range(10):
        p = [rng() for _ in range(1 << k)]
        omega, z = rng(), rng()
        c = pk.commit(p, omega)
        t = T.EvmTranscript()
        I.ipa_create_proof(pk, p, z, omega, t, rng)
        accs.append(I.ipa_succinct_verify(pk.h, pk.s, [(1, c)], z, I.poly_eval(p, z),
                                          I.ipa_read_proof(True, k, T.EvmTranscript(t.finalize()))))
    t = T.EvmTranscript()
    I.ipa_as_create_proof(pk, accs, t, rng)
    acc = I.ipa_as_verify(pk.h, pk.s, accs, I.ipa_as_read_proof(True, k, accs, T.EvmTranscript(t.finalize())))
    assert I.ipa_decide(pk.g, acc)


@pytest.mark.parametrize("idx", range(2))
def test_golden_bgh19_multiopen(idx):
    """`IpaAs<Bgh19>` as a PCS (multiopen/bgh19.rs:26-96) on an honest multi-open proof over six
    polynomials with rotation sets {0}, {0,1}, {0,1,-1}, {1,0}, {-3} and a repeated query."""
    import kzg as K

    c = load_cases("bgh19")[idx]
    g, h, s, coms, x, queries, proof, exp = bgh19_case(c)
    Tr = TR[c["transcript"]]
    sets = K.bdfg21_query_sets(queries)
    assert [len(st["shifts"]) for st in sets] == [1, 2, 3, 1] and sets[1]["polys"] == [1, 3]
    msms = [K.Msm.base(p) for p in coms]
    acc = I.bgh19_verify(g[0], h, s, msms, x, queries, I.bgh19_read_proof(c["k"], queries, Tr(proof)))
    assert (acc[0], acc[1]) == (exp[0], exp[1]) and I.ipa_decide(g, acc)
    for i in (0, 5, 8):  # any wrong evaluation breaks the final opening
        bad = list(queries)
        bad[i] = (bad[i][0], bad[i][1], (bad[i][2] + 1) % O.R)
        with pytest.raises(I.IpaError):
            I.bgh19_verify(g[0], h, s, msms, x, bad, I.bgh19_read_proof(c["k"], bad, Tr(proof)))
    swapped = [msms[1], msms[0]] + msms[2:]
    with pytest.raises(I.IpaError):
        I.bgh19_verify(g[0], h, s, swapped, x, queries, I.bgh19_read_proof(c["k"], queries, Tr(proof)))
    with pytest.raises(T.TranscriptError):
        I.bgh19_read_proof(c["k"], queries, Tr(proof[:-7]))


@pytest.mark.parametrize("kind", ["evm", "poseidon"])
@pytest.mark.parametrize("lin", [None, "MinusVanishingTimesQuotient"])
def test_plonk_over_ipa_forger_and_verifier(kind, lin):
    """`PlonkVerifier<IpaAs<Bgh19>>` (the reference's system/halo2/test/ipa/native.rs) in the oracle:
    a proof forged under a committing key with known discrete logs passes the succinct check AND
    decide; a changed instance fails the succinct check."""
    import plonk as P
    import plonk_synth as S

    rng = random.Random("%s-%s" % (kind, lin))
    k = 4
    pr, dl = S.standard_plonk_protocol(rng, k=k, linearization=lin, num_instance=(3,))
    inst = [[rng.randrange(O.R) for _ in range(3)]]
    kd = {"g": [rng.randrange(1, O.R) for _ in range(1 << k)], "h": rng.randrange(1, O.R), "s": rng.randrange(1, O.R)}
    g = [O.g1_mul(O.G1_GEN, c) for c in kd["g"]]
    h, s = O.g1_mul(O.G1_GEN, kd["h"]), O.g1_mul(O.G1_GEN, kd["s"])
    proof = P.forge_proof_ipa(pr, inst, kd, TR[kind], rng, dl)
    accs = P.succinct_verify_ipa(g[0], h, s, pr, inst, P.plonk_proof_read(pr, inst, TR[kind](proof), "bgh19"))
    assert len(accs) == 1 and len(accs[0][0]) == k and I.ipa_decide(g, accs[0])
    bad = [[inst[0][0], inst[0][1], (inst[0][2] + 1) % O.R]]
    with pytest.raises(I.IpaError):
        P.succinct_verify_ipa(g[0], h, s, pr, bad, P.plonk_proof_read(pr, bad, TR[kind](proof), "bgh19"))
    with pytest.raises(T.TranscriptError):
        P.plonk_proof_read(pr, inst, TR[kind](proof[:-1]), "bgh19")
