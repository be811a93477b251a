"""Generates tests/golden/ipa.json: honest IPA openings + one IpaAs accumulation step per
(transcript, zk) from the Python oracle (oracle/ipa.py = restatement of the reference's
pcs/ipa.rs + pcs/ipa/accumulation.rs; the reference cannot run here).  Run from the repo root:
    python tests/golden/gen_golden_ipa.py
Everything the verifier side needs is in the file (keys, commitments, proof bytes, expected
accumulators), so the tests do not depend on this script's RNG or on the point sampler."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bn254 as O  # noqa: E402
import coracle as C  # noqa: E402
import ipa as I  # noqa: E402
from transcript import EvmTranscript, PoseidonTranscript  # noqa: E402

K, N_OPEN = 5, 3


def hx(b):
    return bytes(b).hex()


def acc_json(acc):
    return {"xi": [hx(O.fe_to_bytes(x)) for x in acc[0]], "u": hx(O.g1_to_bytes(acc[1]))}


def main():
    cases = []
    for tname, T in (("evm", EvmTranscript), ("poseidon", PoseidonTranscript)):
        for zk in (False, True):
            rnd = random.Random("ipa-%s-%d" % (tname, zk))
            rng = lambda: rnd.randrange(O.R)  # noqa: E731
            raw = C.sample_points(rnd.randrange(1 << 32), (1 << K) + 2)
            pts = [O.g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range((1 << K) + 2)]
            pk = I.IpaProvingKey(K, pts[:1 << K], pts[1 << K], pts[(1 << K) + 1] if zk else None)
            openings, accs = [], []
            for _ in range(N_OPEN):
                p = [rng() for _ in range(1 << K)]
                omega = rng() if zk else None
                c, z = pk.commit(p, omega), rng()
                v = I.poly_eval(p, z)
                t = T()
                acc = I.ipa_create_proof(pk, p, z, omega, t, rng)
                proof = t.finalize()
                got = I.ipa_succinct_verify(pk.h, pk.s, [(1, c)], z, v, I.ipa_read_proof(zk, K, T(proof)))
                assert got == acc and I.ipa_decide(pk.g, acc)
                openings.append({"commitment": hx(O.g1_to_bytes(c)), "z": hx(O.fe_to_bytes(z)), "eval": hx(O.fe_to_bytes(v)),
                                 "proof": hx(proof), "accumulator": acc_json(acc)})
                accs.append(acc)
            t = T()
            acc = I.ipa_as_create_proof(pk, accs, t, rng)
            as_proof = t.finalize()
            got = I.ipa_as_verify(pk.h, pk.s, accs, I.ipa_as_read_proof(zk, K, accs, T(as_proof)))
            assert got == acc and I.ipa_decide(pk.g, acc)
            cases.append({"transcript": tname, "zk": zk, "k": K, "g": [hx(O.g1_to_bytes(p)) for p in pk.g],
                          "h": hx(O.g1_to_bytes(pk.h)), "s": hx(O.g1_to_bytes(pk.s)) if zk else None,
                          "openings": openings, "as_proof": hx(as_proof), "as_accumulator": acc_json(acc)})
    with open(os.path.join(ROOT, "tests", "golden", "ipa.json"), "w") as f:
        json.dump({"note": "oracle/ipa.py output (parity unpinned: the reference has no IPA fixtures)", "cases": cases}, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
