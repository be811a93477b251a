"""CPU: the DEVICE math headers (csrc/fq.cuh, g1.cuh, tower.cuh, pairing.cuh)
compiled for the host and checked against the big-integer oracle -- catches
formula bugs before any GPU minute is spent."""
import ctypes
import random

import bn254 as O


def _buf(n):
    return ctypes.create_string_buffer(n)


def test_fq_ops(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(7)
    vals = [0, 1, 2, O.P - 1, O.P - 2] + [rng.randrange(O.P) for _ in range(60)]
    o = _buf(32)
    for a in vals:
        for b in vals[:12]:
            L.ht_fq_mul(O.fe_to_bytes(a), O.fe_to_bytes(b), o)
            assert O.fe_from_bytes(o.raw) == a * b % O.P
            L.ht_fq_add(O.fe_to_bytes(a), O.fe_to_bytes(b), o)
            assert O.fe_from_bytes(o.raw) == (a + b) % O.P
            L.ht_fq_sub(O.fe_to_bytes(a), O.fe_to_bytes(b), o)
            assert O.fe_from_bytes(o.raw) == (a - b) % O.P
    a = rng.randrange(1, O.P)
    L.ht_fq_inv(O.fe_to_bytes(a), o)
    assert O.fe_from_bytes(o.raw) == pow(a, -1, O.P)


def test_g1_group_law_and_exceptional_cases(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(8)
    Pt = O.g1_mul(O.G1_GEN, rng.randrange(O.R))
    Q = O.g1_mul(O.G1_GEN, rng.randrange(O.R))
    o = _buf(64)
    for x, y in [(Pt, Q), (Pt, Pt), (Pt, O.g1_neg(Pt)), (None, Q), (Pt, None), (None, None)]:
        L.ht_g1_add(O.g1_to_bytes(x), O.g1_to_bytes(y), o)
        assert O.g1_from_bytes(o.raw) == O.g1_add(x, y)
    for k in [0, 1, 2, 3, O.R - 1, O.R, rng.randrange(O.R), (1 << 256) - 1]:
        L.ht_g1_mul(O.g1_to_bytes(Pt), int(k).to_bytes(32, "little"), o)
        assert O.g1_from_bytes(o.raw) == O.g1_mul(Pt, k)
    for k1, k2, A, B in [(5, 7, Pt, Q), (5, 5, Pt, Pt), (5, O.R - 5, Pt, Pt), (0, 3, Pt, Q), (3, 0, Pt, Q)]:
        L.ht_g1_lincomb(O.g1_to_bytes(A), k1.to_bytes(32, "little"), O.g1_to_bytes(B), k2.to_bytes(32, "little"), o)
        assert O.g1_from_bytes(o.raw) == O.g1_add(O.g1_mul(A, k1), O.g1_mul(B, k2))
    assert L.ht_g1_on_curve(O.g1_to_bytes(Pt)) == 1
    assert L.ht_g1_on_curve(O.g1_to_bytes((Pt[0], (Pt[1] + 1) % O.P))) == 0


def _rfq12(rng):
    f2 = lambda: O.Fq2(rng.randrange(O.P), rng.randrange(O.P))  # noqa: E731
    return O.Fq12(O.Fq6(f2(), f2(), f2()), O.Fq6(f2(), f2(), f2()))


def test_fq12_tower(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(9)
    A, B = _rfq12(rng), _rfq12(rng)
    o = _buf(384)
    L.ht_fq12_mul(A.to_bytes(), B.to_bytes(), o)
    assert o.raw == (A * B).to_bytes()
    L.ht_fq12_sqr(A.to_bytes(), o)
    assert o.raw == (A * A).to_bytes()
    L.ht_fq12_inv(A.to_bytes(), o)
    assert o.raw == A.inv().to_bytes()
    for k in (1, 2, 3):
        L.ht_fq12_frob(A.to_bytes(), k, o)
        assert o.raw == A.pow(O.P ** k).to_bytes()


def test_final_exponentiation_is_exact(hosttest_lib):
    rng = random.Random(10)
    A = _rfq12(rng)
    o = _buf(384)
    hosttest_lib.ht_final_exp(A.to_bytes(), o)
    assert o.raw == O.final_exponentiation(A).to_bytes()


def test_pairing_matches_oracle_value(hosttest_lib, golden_decider):
    L = hosttest_lib
    rng = random.Random(11)
    a, b = rng.randrange(O.R), rng.randrange(O.R)
    Pa, Qb = O.g1_mul(O.G1_GEN, a), O.g2_mul(O.G2_GEN, b)
    o = _buf(384)
    L.ht_pairing_product(O.g1_to_bytes(Pa), O.g2_to_bytes(Qb), 1, o)
    assert o.raw == O.pairing(Pa, Qb).to_bytes()
    g2 = bytes.fromhex(golden_decider["g2"])
    ns_g2 = O.g2_to_bytes(O.g2_neg(O.g2_from_bytes(bytes.fromhex(golden_decider["s_g2"]))))
    for case in golden_decider["cases"]:
        acc = bytes.fromhex(case["acc"])
        L.ht_pairing_product(acc, g2 + ns_g2, 2, o)
        assert o.raw == bytes.fromhex(case["gt"]), case["name"]
        assert (o.raw == O.FQ12_ONE.to_bytes()) == case["accept"]
