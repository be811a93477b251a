"""Generates tests/golden/plonk_forged.json: StandardPlonk-shaped protocols with
proofs forged under the toy SRS secret (oracle/plonk.py), and the accumulators
the succinct verifier must output.  Run from the repo root:
    python tests/golden/gen_golden_plonk.py
Data only (protocol bytes in the tests' wire format, instances, proof bytes,
expected accumulators)."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bn254 as O
import plonk as P
import plonk_synth as S
import transcript as T

SECRET = 0x1F2E3D4C5B6A79887766554433221100AABBCCDDEEFF


def main():
    rng = random.Random(0x910C)
    cases = []
    for name, mos, kind, lin in (("gwc19_evm", "gwc19", 0, None), ("bdfg21_poseidon", "bdfg21", 1, None),
                                 ("gwc19_poseidon_linearized", "gwc19", 1, "MinusVanishingTimesQuotient")):
        pr, dl = S.standard_plonk_protocol(rng, linearization=lin)
        inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
        mk = (lambda: T.EvmTranscript()) if kind == 0 else (lambda: T.PoseidonTranscript())
        proof = P.forge_proof(pr, inst, SECRET, mk, mos, rng, dl)
        t = T.EvmTranscript(proof) if kind == 0 else T.PoseidonTranscript(proof)
        pf = P.plonk_proof_read(pr, inst, t, mos)
        accs = P.succinct_verify(O.G1_GEN, pr, inst, pf, mos)
        assert accs[0][0] == O.g1_mul(accs[0][1], SECRET)
        cases.append({"name": name, "mos": mos, "transcript": kind, "protocol": S.pack_protocol(pr).hex(),
                      "instances": S.pack_instances(inst).hex(), "proof": proof.hex(),
                      "z": hex(pf["z"]), "challenges": [hex(c) for c in pf["challenges"]],
                      "accumulators": [(O.g1_to_bytes(a) + O.g1_to_bytes(b)).hex() for a, b in accs]})
    with open(os.path.join(ROOT, "tests", "golden", "plonk_forged.json"), "w") as f:
        json.dump({"secret": hex(SECRET), "cases": cases}, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
