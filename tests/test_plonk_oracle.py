"""CPU: the oracle restatement of the PLONK succinct verifier (oracle/plonk.py) --
expression evaluation, proof reading order, commitments/queries, both multi-open
schemes, all three linearization strategies -- checked by the property that
proofs forged under a toy SRS verify (lhs = s * rhs) and perturbed ones do not,
and against the committed fixture tests/golden/plonk_forged.json."""
import json
import os
import random

import pytest

import bn254 as O
import plonk as P
import plonk_synth as S
import transcript as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECRET = 0x1F2E3D4C5B6A79887766554433221100AABBCCDDEEFF


def _t(kind, proof=None):
    cls = T.EvmTranscript if kind == 0 else T.PoseidonTranscript
    return cls() if proof is None else cls(proof)


@pytest.mark.parametrize("mos", ["gwc19", "bdfg21"])
@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("lin", [None, "WithoutConstant", "MinusVanishingTimesQuotient"])
def test_forged_proofs_verify_in_the_oracle(mos, kind, lin):
    rng = random.Random(hash((mos, kind, lin)) & 0xFFFF)
    pr, dl = S.standard_plonk_protocol(rng, linearization=lin)
    inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
    proof = P.forge_proof(pr, inst, SECRET, lambda: _t(kind), mos, rng, dl)
    t = _t(kind, proof)
    pf = P.plonk_proof_read(pr, inst, t, mos)
    assert t.pos == len(t.stream)
    lhs, rhs = P.succinct_verify(O.G1_GEN, pr, inst, pf, mos)[0]
    assert lhs == O.g1_mul(rhs, SECRET)
    bad = bytearray(proof)
    bad[len(bad) // 3] ^= 0x10
    try:
        t = _t(kind, bytes(bad))
        pf = P.plonk_proof_read(pr, inst, t, mos)
        lhs, rhs = P.succinct_verify(O.G1_GEN, pr, inst, pf, mos)[0]
        assert lhs != O.g1_mul(rhs, SECRET)
    except T.TranscriptError:
        pass


def test_expression_evaluation_order_and_helpers():
    # DistributePowers([a, b, c], y) = (a*y + b)*y + c  (protocol.rs:358-370)
    e = ("dpow", [("const", 2), ("const", 3), ("const", 5)], ("const", 7))
    ev = lambda x: P.expr_evaluate(x, lambda s: s, None, None, None, lambda a: -a % O.R, lambda a, b: (a + b) % O.R,
                                   lambda a, b: a * b % O.R, lambda a, s: a * s % O.R)
    assert ev(e) == (2 * 7 + 3) * 7 + 5
    assert ev(("dpow", [("const", 9)], ("const", 7))) == 9
    ex = ("sum", ("prod", ("lagrange", -3), ("poly", 4, 1)), ("scaled", ("neg", ("lagrange", 0)), 5))
    assert P.used_lagrange(ex) == {-3, 0}
    assert P.used_query(ex) == {(4, 1)}


def test_lagrange_range_for_instance_evaluation():
    """`langranges()` (protocol.rs:80-111): instance polys queried at rotations
    [min, max] need l_i for i in [-max, max_len + |min|)."""
    rng = random.Random(3)
    pr, _ = S.standard_plonk_protocol(rng, num_instance=(2, 3))
    got = P.protocol_lagranges(pr)
    # numerator uses l_0, l_last=-6, l_blind=-5..-1 and instance polys at rotations 0 and +1
    assert got == set(range(-6, 1)) | set(range(-1, 3))
    dom = pr["domain"]
    z = rng.randrange(O.R)
    cpe = P.CommonPolyEval(dom, got, z)
    # l_i(z) = (z^n - 1) / (n (z/w^i - 1))  against the closed form
    for i in got:
        w = dom.rotate_scalar(1, i)
        assert cpe.get(("lagrange", i)) == (pow(z, dom.n, O.R) - 1) * w % O.R * pow(dom.n * (z - w) % O.R, -1, O.R) % O.R
    assert pow(dom.gen, dom.n, O.R) == 1 and pow(dom.gen, dom.n // 2, O.R) == O.R - 1


def test_golden_fixture_reproduces():
    with open(os.path.join(ROOT, "tests", "golden", "plonk_forged.json")) as f:
        g = json.load(f)
    assert int(g["secret"], 16) == SECRET
    rng = random.Random(0x910C)  # the generator's seed: the protocol objects are rebuilt, then compared byte for byte
    for case, (mos, kind, lin) in zip(g["cases"], (("gwc19", 0, None), ("bdfg21", 1, None),
                                                   ("gwc19", 1, "MinusVanishingTimesQuotient"))):
        pr, dl = S.standard_plonk_protocol(rng, linearization=lin)
        inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
        proof = P.forge_proof(pr, inst, SECRET, lambda: _t(kind), mos, rng, dl)
        assert S.pack_protocol(pr).hex() == case["protocol"]
        assert proof.hex() == case["proof"]
        pf = P.plonk_proof_read(pr, inst, _t(kind, bytes.fromhex(case["proof"])), mos)
        assert hex(pf["z"]) == case["z"]
        accs = P.succinct_verify(O.G1_GEN, pr, inst, pf, mos)
        assert [(O.g1_to_bytes(a) + O.g1_to_bytes(b)).hex() for a, b in accs] == case["accumulators"]


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_protocol_shapes_verify_in_the_oracle(seed):
    """The forger + oracle verifier on random protocol shapes (the same generator the GPU
    fuzz test drives the C++ mirror with)."""
    rng = random.Random(7000 + seed)
    lin = rng.choice([None, None, "WithoutConstant", "MinusVanishingTimesQuotient"])
    mos = rng.choice(["gwc19", "bdfg21"])
    kind = rng.randrange(2)
    pr, dl = S.random_protocol(rng, lin)
    inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
    proof = P.forge_proof(pr, inst, SECRET, lambda: _t(kind), mos, rng, dl)
    t = _t(kind, proof)
    pf = P.plonk_proof_read(pr, inst, t, mos)
    assert t.pos == len(t.stream)
    lhs, rhs = P.succinct_verify(O.G1_GEN, pr, inst, pf, mos)[0]
    assert lhs == O.g1_mul(rhs, SECRET)
