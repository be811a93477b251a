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
import kzg as K  # noqa: E402
from transcript import EvmTranscript, PoseidonTranscript  # noqa: E402

K_IPA, N_OPEN = 5, 3


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
            raw = C.sample_points(rnd.randrange(1 << 32), (1 << K_IPA) + 2)
            pts = [O.g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range((1 << K_IPA) + 2)]
            pk = I.IpaProvingKey(K_IPA, pts[:1 << K_IPA], pts[1 << K_IPA], pts[(1 << K_IPA) + 1] if zk else None)
            openings, accs = [], []
            for _ in range(N_OPEN):
                p = [rng() for _ in range(1 << K_IPA)]
                omega = rng() if zk else None
                c, z = pk.commit(p, omega), rng()
                v = I.poly_eval(p, z)
                t = T()
                acc = I.ipa_create_proof(pk, p, z, omega, t, rng)
                proof = t.finalize()
                got = I.ipa_succinct_verify(pk.h, pk.s, [(1, c)], z, v, I.ipa_read_proof(zk, K_IPA, T(proof)))
                assert got == acc and I.ipa_decide(pk.g, acc)
                openings.append({"commitment": hx(O.g1_to_bytes(c)), "z": hx(O.fe_to_bytes(z)), "eval": hx(O.fe_to_bytes(v)),
                                 "proof": hx(proof), "accumulator": acc_json(acc)})
                accs.append(acc)
            t = T()
            acc = I.ipa_as_create_proof(pk, accs, t, rng)
            as_proof = t.finalize()
            got = I.ipa_as_verify(pk.h, pk.s, accs, I.ipa_as_read_proof(zk, K_IPA, accs, T(as_proof)))
            assert got == acc and I.ipa_decide(pk.g, acc)
            cases.append({"transcript": tname, "zk": zk, "k": K_IPA, "g": [hx(O.g1_to_bytes(p)) for p in pk.g],
                          "h": hx(O.g1_to_bytes(pk.h)), "s": hx(O.g1_to_bytes(pk.s)) if zk else None,
                          "openings": openings, "as_proof": hx(as_proof), "as_accumulator": acc_json(acc)})
    bgh = []
    for tname, T in (("evm", EvmTranscript), ("poseidon", PoseidonTranscript)):
        rnd = random.Random("bgh19-%s" % tname)
        rng = lambda: rnd.randrange(O.R)  # noqa: E731
        k = 4
        n = 1 << k
        raw = C.sample_points(rnd.randrange(1 << 32), n + 2)
        pts = [O.g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range(n + 2)]
        pk = I.IpaProvingKey(k, pts[:n], pts[n], pts[n + 1])
        polys = [[rng() for _ in range(n)] for _ in range(6)]
        blinds = [rng() for _ in polys]
        coms = [pk.commit(p, b) for p, b in zip(polys, blinds)]
        x = rng()
        w = pow(5, (O.R - 1) // n, O.R)
        shifts = [1, w, pow(w, O.R - 2, O.R), pow(w, n - 3, O.R)]
        # rotation sets {0}, {0,1}, {0,1,-1}, {1,0} (= {0,1}: joins it), {-3}; one repeated query
        spec = [(0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2), (3, 1), (3, 0), (4, 3), (5, 0), (0, 0)]
        queries = [(p, shifts[s], I.poly_eval(polys[p], x * shifts[s] % O.R)) for p, s in spec]
        t = T()
        I.bgh19_create_proof(pk, polys, blinds, x, queries, t, rng)
        proof = t.finalize()
        acc = I.bgh19_verify(pk.g[0], pk.h, pk.s, [K.Msm.base(c) for c in coms], x, queries,
                             I.bgh19_read_proof(k, queries, T(proof)))
        assert I.ipa_decide(pk.g, acc)
        bgh.append({"transcript": tname, "k": k, "g": [hx(O.g1_to_bytes(p)) for p in pk.g], "h": hx(O.g1_to_bytes(pk.h)),
                    "s": hx(O.g1_to_bytes(pk.s)), "commitments": [hx(O.g1_to_bytes(c)) for c in coms],
                    "x": hx(O.fe_to_bytes(x)),
                    "queries": [[p, hx(O.fe_to_bytes(sh)), hx(O.fe_to_bytes(ev))] for p, sh, ev in queries],
                    "proof": hx(proof), "accumulator": acc_json(acc)})
    with open(os.path.join(ROOT, "tests", "golden", "ipa.json"), "w") as f:
        json.dump({"note": "oracle/ipa.py output (parity unpinned: the reference has no IPA fixtures)", "cases": cases,
                   "bgh19": bgh}, f)
    print("wrote", len(cases), "cases +", len(bgh), "bgh19")


if __name__ == "__main__":
    main()
