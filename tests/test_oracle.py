"""CPU: the oracle itself -- public constants, algebraic invariants, golden
vectors, Python big-int vs the C restatement.  (No reference KATs exist for
this path: SURVEY.md section 4.)"""
import random

import bn254 as O
import coracle as C


def test_public_constants():
    assert O.g1_double(O.G1_GEN) == (
        1368015179489954701390400359078579693043519447331113978918064868415326638035,
        9918110051302171585080402603319702774565515993150576347155970296011118125764,
    )
    assert O.g1_mul(O.G1_GEN, O.R) is None
    assert O.g2_is_on_curve(O.G2_GEN) and O.g2_mul(O.G2_GEN, O.R) is None
    assert O.pippenger_window_size(1 << 20) == 16 and O.pippenger_window_size(1 << 24) == 19


def test_pairing_bilinear_nondegenerate():
    e = O.pairing(O.G1_GEN, O.G2_GEN)
    assert not e.is_one() and e.pow(O.R).is_one()
    a, b = 0xC0FFEE, 0xBADC0DE
    assert O.pairing(O.g1_mul(O.G1_GEN, a), O.g2_mul(O.G2_GEN, b)) == e.pow(a * b)


def test_golden_msm_python_and_c(golden_msm):
    for case in golden_msm:
        s, p, exp = bytes.fromhex(case["scalars"]), bytes.fromhex(case["points"]), bytes.fromhex(case["expected"])
        assert C.msm_naive(s, p) == exp, case["name"]
        assert C.msm_pippenger(s, p, 1) == exp, case["name"]
        assert C.msm_pippenger(s, p, 3) == exp, case["name"]
        n = len(s) // 32
        if n <= 65:
            sc = [int.from_bytes(s[32 * i:32 * i + 32], "little") for i in range(n)]
            pts = [O.g1_from_bytes(p[64 * i:64 * i + 64]) for i in range(n)]
            assert O.g1_to_bytes(O.g1_msm_pippenger(sc, pts)) == exp, case["name"]
            assert O.g1_to_bytes(O.g1_msm_naive(sc, pts)) == exp, case["name"]


def test_golden_decider_python(golden_decider):
    g2 = O.g2_from_bytes(bytes.fromhex(golden_decider["g2"]))
    s_g2 = O.g2_from_bytes(bytes.fromhex(golden_decider["s_g2"]))
    for case in golden_decider["cases"][:3]:
        acc = bytes.fromhex(case["acc"])
        lhs, rhs = O.g1_from_bytes(acc[:64]), O.g1_from_bytes(acc[64:])
        assert O.kzg_decide(lhs, rhs, g2, s_g2) == case["accept"], case["name"]


def test_empty_msm_panics_like_reference():
    import pytest

    with pytest.raises(ValueError):
        O.g1_msm_naive([], [])
    with pytest.raises(ValueError):
        O.g1_msm_pippenger([], [])
    with pytest.raises(ValueError):
        C.msm_naive(b"", b"")
    with pytest.raises(ValueError):
        C.msm_pippenger(b"", b"")


def test_c_oracle_linearity_and_sampler():
    n = 300
    s, p = C.sample_scalars(11, n), C.sample_points(12, n)
    assert all(C.g1_is_on_curve(p[64 * i:64 * i + 64]) for i in range(n))
    assert all(int.from_bytes(s[32 * i:32 * i + 32], "little") < O.R for i in range(n))
    full = C.msm_pippenger(s, p, 1)
    h = 123
    a = C.msm_pippenger(s[:32 * h], p[:64 * h], 1)
    b = C.msm_naive(s[32 * h:], p[64 * h:])
    assert C.g1_add(a, b) == full
    # first/offset form of the sampler is a pure function of the element index
    assert C.sample_scalars(11, 10, first=5) == s[32 * 5:32 * 15]
    assert C.sample_points(12, 10, first=5) == p[64 * 5:64 * 15]


def test_python_vs_c_random():
    rng = random.Random(99)
    for n in (1, 5, 17):
        sc = [rng.randrange(O.R) for _ in range(n)]
        pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(n)]
        s = b"".join(O.fe_to_bytes(x) for x in sc)
        p = b"".join(O.g1_to_bytes(x) for x in pts)
        exp = O.g1_to_bytes(O.g1_msm_naive(sc, pts))
        assert C.msm_naive(s, p) == exp and C.msm_pippenger(s, p, 2) == exp


def test_golden_kzg_layer_oracle_reproduces():
    """The committed KZG-layer fixture is what oracle/kzg.py computes today, and
    every accumulator in it is a valid opening under the toy SRS."""
    import json
    import os

    import kzg as K

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", "kzg_layer.json")))
    secret = int(g["secret"], 16)
    ka = g["kzg_as"]
    raw = bytes.fromhex(ka["accumulators"])
    accs = [(O.g1_from_bytes(raw[128 * i:128 * i + 64]), O.g1_from_bytes(raw[128 * i + 64:128 * i + 128]))
            for i in range(len(raw) // 128)]
    for lhs, rhs in accs:
        assert lhs == O.g1_mul(rhs, secret)
    r = O.fe_from_bytes(bytes.fromhex(ka["r"]))
    res = K.kzg_as_verify(accs, r)
    assert (O.g1_to_bytes(res[0]) + O.g1_to_bytes(res[1])).hex() == ka["result"]
    lim = bytes.fromhex(g["limbs"]["limbs"])
    limbs = [O.fe_from_bytes(lim[32 * i:32 * i + 32]) for i in range(16)]
    back = K.limbs_from_repr(limbs)
    assert (O.g1_to_bytes(back[0]) + O.g1_to_bytes(back[1])).hex() == g["limbs"]["accumulator"]


def test_golden_decider_c_oracle_gt_bytes(golden_decider):
    """The C restatement of the pairing decider (oracle/c/bn254_pairing.inc <- pcs/kzg/decider.rs:70-93: prepared
    G2 lines, signed-digit Miller loop, cyclotomic final exponentiation) produces the SAME Gt element, byte for
    byte, as the independent big-integer oracle (affine lines, plain f^((p^12-1)/r)) on every golden case."""
    g2, s_g2 = bytes.fromhex(golden_decider["g2"]), bytes.fromhex(golden_decider["s_g2"])
    accs = b""
    for case in golden_decider["cases"]:
        acc = bytes.fromhex(case["acc"])
        accs += acc
        assert C.kzg_decide(g2, s_g2, acc) == case["accept"], case["name"]
        if case.get("gt"):
            assert C.kzg_pairing_value(g2, s_g2, acc).hex() == case["gt"], case["name"]
    exp = [c["accept"] for c in golden_decider["cases"]]
    for threads in (1, 3):
        allok, oks = C.kzg_decide_all(g2, s_g2, accs, threads)
        assert oks == exp and allok == all(exp)
    assert C.kzg_decide_all(g2, s_g2, b"", 1) == (True, [])


def test_c_pairing_random_vs_python_and_bilinear():
    import random

    rng = random.Random(0xDEC1DE)
    assert C.lib().oracle_selftest_cyclotomic(O.g2_to_bytes(O.G2_GEN), O.g1_to_bytes(O.G1_GEN)) == 1
    for _ in range(3):
        a, b, s = (rng.randrange(1, O.R) for _ in range(3))
        g2, s_g2 = O.g2_mul(O.G2_GEN, b), O.g2_mul(O.G2_GEN, b * s % O.R)
        lhs, rhs = O.g1_mul(O.G1_GEN, a), O.g1_mul(O.G1_GEN, rng.randrange(1, O.R))
        f = O.final_exponentiation(O.miller_loop([(lhs, g2), (rhs, O.g2_neg(s_g2))]))
        acc = O.g1_to_bytes(lhs) + O.g1_to_bytes(rhs)
        assert C.kzg_pairing_value(O.g2_to_bytes(g2), O.g2_to_bytes(s_g2), acc) == f.to_bytes()
        # bilinearity through the decider: e(s a G, b G2) e(a G, -(s b) G2) = 1
        good = O.g1_to_bytes(O.g1_mul(O.G1_GEN, a * s % O.R)) + O.g1_to_bytes(lhs)
        assert C.kzg_decide(O.g2_to_bytes(g2), O.g2_to_bytes(s_g2), good)
    # identity members contribute 1 (multi_miller_loop skips them)
    z = bytes(64)
    assert C.kzg_decide(O.g2_to_bytes(O.G2_GEN), O.g2_to_bytes(O.G2_GEN), z + z)
    assert not C.kzg_decide(O.g2_to_bytes(O.G2_GEN), O.g2_to_bytes(O.G2_GEN), O.g1_to_bytes(O.G1_GEN) + z)
