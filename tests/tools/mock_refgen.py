"""MOCK of tools/refgen for plumbing tests: writes files with refgen's schema and names into the directory given as
argv[1], computed by the ORACLE (oracle/*.py), not by the reference.  They pin nothing; they only keep
tests/test_reference_vectors.py's loaders exercised until real vectors exist.  Every file says so in "generator"."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bn254 as O
import coracle as C
import kzg as K
import transcript as T

MOCK = "tests/tools/mock_refgen.py (MOCK: oracle output, NOT the reference)"


def accb(a):
    return O.g1_to_bytes(a[0]) + O.g1_to_bytes(a[1])


def main(out):
    rng = random.Random(0x5EED0001)
    os.makedirs(out, exist_ok=True)
    cases = []
    for n in (1, 2, 3, 21, 64, 65, 1024):
        s, p = C.sample_scalars(n, n), C.sample_points(n + 1, n)
        cases.append({"name": "native_loader_n%d" % n, "scalars": s.hex(), "points": p.hex(), "expected": C.msm_naive(s, p).hex()})
    s, p = C.sample_scalars(7, 1 << 10), C.sample_points(8, 1 << 10)
    cases.append({"name": "util_msm_2p10", "scalars": s.hex(), "points": p.hex(), "expected": C.msm_pippenger(s, p, 1).hex()})
    json.dump({"generator": MOCK, "cases": cases}, open(os.path.join(out, "ref_g1_msm.json"), "w"))
    sec = rng.randrange(1, O.R)
    s_g2 = O.g2_mul(O.G2_GEN, sec)

    def valid():
        rhs = O.g1_mul(O.G1_GEN, rng.randrange(1, O.R))
        return (O.g1_mul(rhs, sec), rhs)

    accs = [valid() for _ in range(8)]
    t = T.EvmTranscript()
    for lhs, rhs in accs:
        t.common_ec_point(lhs)
        t.common_ec_point(rhs)
    folded = K.kzg_as_verify(accs, t.squeeze_challenge())
    json.dump({"generator": MOCK, "accumulators": b"".join(accb(a) for a in accs).hex(), "result": accb(folded).hex()},
              open(os.path.join(out, "ref_kzg_as.json"), "w"))
    tp = T.PoseidonTranscript()
    for lhs, rhs in accs:
        tp.common_ec_point(lhs)
        tp.common_ec_point(rhs)
    folded_p = K.kzg_as_verify(accs, tp.squeeze_challenge())
    json.dump({"generator": MOCK, "accumulators": b"".join(accb(a) for a in accs).hex(), "result": accb(folded_p).hex()},
              open(os.path.join(out, "ref_kzg_as_poseidon.json"), "w"))
    dc = [{"name": "valid_0", "acc": accb(accs[0]).hex(), "accept": True},
          {"name": "folded", "acc": accb(folded).hex(), "accept": True},
          {"name": "invalid_0", "acc": accb((O.g1_add(accs[1][0], O.G1_GEN), accs[1][1])).hex(), "accept": False}]
    json.dump({"generator": MOCK, "g1": O.g1_to_bytes(O.G1_GEN).hex(), "g2": O.g2_to_bytes(O.G2_GEN).hex(),
               "s_g2": O.g2_to_bytes(s_g2).hex(), "cases": dc}, open(os.path.join(out, "ref_kzg_decider.json"), "w"))
    json.dump({"generator": MOCK, "accumulator": accb(folded).hex(),
               "limbs": b"".join(O.fe_to_bytes(x) for x in K.accumulator_to_limbs(folded)).hex()},
              open(os.path.join(out, "ref_limbs.json"), "w"))
    import interchange_fmt as X
    import plonk_synth as S

    pr, _ = S.standard_plonk_protocol(random.Random(5))
    inst = [[rng.randrange(O.R) for _ in range(m)] for m in pr["num_instance"]]
    proof = bytes(rng.randrange(256) for _ in range(64))
    open(os.path.join(out, "ref_snark.bin"), "wb").write(X.snark_to_bincode(pr, inst, proof))
    open(os.path.join(out, "ref_snark.json"), "wb").write(X.snark_to_json(pr, inst, proof))


if __name__ == "__main__":
    main(sys.argv[1])
