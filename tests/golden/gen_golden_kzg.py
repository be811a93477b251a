#!/usr/bin/env python3
"""Golden vectors for the KZG accumulation layer (tests/golden/kzg_layer.json),
generated from oracle/kzg.py + oracle/bn254.py (big-integer restatements).
Shape-faithful synthetic StandardPlonk openings under a toy SRS secret: every
accumulator is VALID by construction.  PARITY UNPINNED against
the reference itself (it cannot run here); see oracle/README.md.

    python tests/golden/gen_golden_kzg.py
"""
import json
import os
import random
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bn254 as O  # noqa: E402
import kzg as K  # noqa: E402
from hostfmt import pack_bdfg21, pack_gwc19  # noqa: E402

SECRET = 0x1F2E3D4C5B6A79887766554433221100AABBCCDDEEFF


def acc_hex(acc):
    return (O.g1_to_bytes(acc[0]) + O.g1_to_bytes(acc[1])).hex()


def main():
    rng = random.Random(0x5EED0003)
    out = {"generator": "tests/golden/gen_golden_kzg.py", "oracle": "oracle/kzg.py", "secret": hex(SECRET),
           "g2": O.g2_to_bytes(O.G2_GEN).hex(), "s_g2": O.g2_to_bytes(O.g2_mul(O.G2_GEN, SECRET)).hex(),
           "gwc19": [], "bdfg21": []}
    accs = []
    for _ in range(3):
        inst = K.synth_gwc19_instance(rng, SECRET)
        acc = K.gwc19_verify(inst["g"], inst["commitments"], inst["z"], inst["queries"], inst["v"], inst["ws"], inst["u"])
        assert acc[0] == O.g1_mul(acc[1], SECRET)
        out["gwc19"].append({"input": pack_gwc19(inst).hex(), "accumulator": acc_hex(acc), "msm_sizes": [21, 3]})
        accs.append(acc)
    for _ in range(3):
        inst = K.synth_bdfg21_instance(rng, SECRET)
        acc = K.bdfg21_verify(inst["g"], inst["commitments"], inst["z"], inst["queries"], inst["mu"], inst["gamma"],
                              inst["w"], inst["z_prime"], inst["w_prime"])
        assert acc[0] == O.g1_mul(acc[1], SECRET)
        out["bdfg21"].append({"input": pack_bdfg21(inst).hex(), "accumulator": acc_hex(acc), "msm_sizes": [20, 1]})
        accs.append(acc)
    r = rng.randrange(O.R)
    b = rng.randrange(O.R)
    blind = (O.g1_mul(O.G1_GEN, SECRET * b % O.R), O.g1_mul(O.G1_GEN, b))
    out["kzg_as"] = {
        "accumulators": "".join(acc_hex(a) for a in accs),
        "r": O.fe_to_bytes(r).hex(),
        "result": acc_hex(K.kzg_as_verify(accs, r)),
        "blind_scalar": O.fe_to_bytes(b).hex(),
        "blind": acc_hex(blind),
        "result_zk": acc_hex(K.kzg_as_verify(accs, r, blind)),
    }
    final = K.kzg_as_verify(accs, r)
    out["limbs"] = {"accumulator": acc_hex(final),
                    "limbs": "".join(O.fe_to_bytes(x).hex() for x in K.accumulator_to_limbs(final))}
    assert O.kzg_decide(final[0], final[1], O.G2_GEN, O.g2_mul(O.G2_GEN, SECRET))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kzg_layer.json"), "w") as f:
        json.dump(out, f)
    print("wrote kzg_layer.json")


if __name__ == "__main__":
    main()
