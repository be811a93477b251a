#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the
big-integer oracle `oracle/bn254.py` (pure math; independent of the C oracle
and of the HIP code).  The reference holds no known-answer data for this path
("zero golden vectors"), and it cannot be run here, so
these vectors are oracle-generated and PARITY UNPINNED against halo2curves;
they pin (a) the C restatement, (b) the HIP kernels and (c) regressions.

    python tests/golden/gen_golden.py
"""
import json
import os
import random
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bn254 as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rand_point(rng):
    while True:
        x = rng.randrange(O.P)
        rhs = (x * x * x + 3) % O.P
        y = pow(rhs, (O.P + 1) // 4, O.P)
        if y * y % O.P == rhs:
            return (x, y if rng.random() < 0.5 else O.P - y)


def hexs(bs):
    return bs.hex()


def msm_case(name, scalars, points):
    exp = O.g1_msm_naive(scalars, points)
    return {
        "name": name,
        "scalars": hexs(b"".join(O.fe_to_bytes(s) for s in scalars)),
        "points": hexs(b"".join(O.g1_to_bytes(p) for p in points)),
        "expected": hexs(O.g1_to_bytes(exp)),
    }


def main():
    rng = random.Random(0x5EED0001)
    cases = []
    for n in (1, 2, 3, 21, 64, 65, 1024):
        sc = [rng.randrange(O.R) for _ in range(n)]
        pts = [rand_point(rng) for _ in range(n)]
        cases.append(msm_case("random_n%d" % n, sc, pts))
    # adversarial sets: legal inputs that hit the exceptional
    # cases of the group law
    p0, p1 = rand_point(rng), rand_point(rng)
    k = rng.randrange(O.R)
    cases.append(msm_case("all_equal_points", [rng.randrange(O.R) for _ in range(8)], [p0] * 8))
    cases.append(msm_case("same_point_same_scalar", [k, k, k], [p0, p0, p0]))
    cases.append(msm_case("p_and_neg_p_cancel", [k, k], [p0, O.g1_neg(p0)]))
    cases.append(msm_case("p_and_neg_p_mixed", [k, k, 5], [p0, O.g1_neg(p0), p1]))
    cases.append(msm_case("zero_scalars", [0, 0, 0], [p0, p1, rand_point(rng)]))
    cases.append(msm_case("zero_and_one", [0, 1, 0], [p0, p1, rand_point(rng)]))
    cases.append(msm_case("scalar_r_minus_1", [O.R - 1, O.R - 1], [p0, p1]))
    cases.append(msm_case("identity_points", [k, 7, 9], [None, p1, None]))
    cases.append(msm_case("all_identity", [k, 7], [None, None]))
    cases.append(msm_case("generator_two", [2], [O.G1_GEN]))
    cases.append(msm_case("small_scalars", list(range(1, 34)), [rand_point(rng) for _ in range(33)]))
    cases.append(msm_case("top_bits", [(1 << 253) + 1, (1 << 253), O.R - 2], [p0, p1, rand_point(rng)]))
    with open(os.path.join(OUT, "g1_msm.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_golden.py", "oracle": "oracle/bn254.py", "cases": cases}, f)

    # KZG decider: toy SRS secret s; accumulator (s*a*G, a*G) is valid
    s = rng.randrange(O.R)
    g2, s_g2 = O.G2_GEN, O.g2_mul(O.G2_GEN, s)
    dec = {"g1": hexs(O.g1_to_bytes(O.G1_GEN)), "g2": hexs(O.g2_to_bytes(g2)), "s_g2": hexs(O.g2_to_bytes(s_g2)),
           "secret": hex(s), "cases": []}
    for i in range(4):
        a = rng.randrange(O.R)
        rhs = O.g1_mul(O.G1_GEN, a)
        lhs = O.g1_mul(rhs, s)
        for label, l, r_ in (("valid", lhs, rhs), ("perturbed_lhs", O.g1_add(lhs, O.G1_GEN), rhs)):
            if i >= 2 and label != "valid":
                continue
            f = O.final_exponentiation(O.miller_loop([(l, g2), (r_, O.g2_neg(s_g2))]))
            dec["cases"].append({
                "name": "%s_%d" % (label, i),
                "acc": hexs(O.g1_to_bytes(l) + O.g1_to_bytes(r_)),
                "accept": f.is_one(),
                "gt": hexs(f.to_bytes()),
            })
    # identities: e(O, g2) * e(O, -s g2) = 1 ; (lhs, O) rejects unless lhs = O
    f = O.final_exponentiation(O.miller_loop([(None, g2), (None, O.g2_neg(s_g2))]))
    dec["cases"].append({"name": "both_identity", "acc": hexs(b"\x00" * 128), "accept": f.is_one(), "gt": hexs(f.to_bytes())})
    l = O.g1_mul(O.G1_GEN, 77)
    f = O.final_exponentiation(O.miller_loop([(l, g2), (None, O.g2_neg(s_g2))]))
    dec["cases"].append({"name": "rhs_identity", "acc": hexs(O.g1_to_bytes(l) + b"\x00" * 64), "accept": f.is_one(), "gt": hexs(f.to_bytes())})
    with open(os.path.join(OUT, "kzg_decider.json"), "w") as f_:
        json.dump({"generator": "tests/golden/gen_golden.py", "oracle": "oracle/bn254.py", **dec}, f_)
    print("wrote", len(cases), "msm cases and", len(dec["cases"]), "decider cases")


if __name__ == "__main__":
    main()
