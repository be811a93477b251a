"""Generates tests/golden/evm_transcript.json from the oracle (oracle/transcript.py).
Run from the repo root:  python tests/golden/gen_golden_transcript.py
Each case: a script of transcript operations (see test_driver.cpp
hd_evm_transcript_script), the proof bytes it reads, and the expected output
bytes / return code.  Data only; the script itself is the provenance."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bn254 as O
import transcript as T


def run_script(ops, proof, kind=0):
    """ops: list of (op, payload).  Returns (rc, out_bytes) exactly as the C++ driver does.
    kind 0 = EvmTranscript, 1 = PoseidonTranscript."""
    t = T.EvmTranscript(proof) if kind == 0 else T.PoseidonTranscript(proof)
    out = b""
    for i, (op, payload) in enumerate(ops):
        try:
            if op == 1:
                out += O.fe_to_bytes(t.squeeze_challenge())
            elif op == 2:
                t.common_scalar(payload)
            elif op == 3:
                t.common_ec_point(payload)
            elif op == 4:
                out += O.fe_to_bytes(t.read_scalar())
            elif op == 5:
                out += O.g1_to_bytes(t.read_ec_point())
            elif op == 6:
                t.write_scalar(payload)
            elif op == 7:
                t.write_ec_point(payload)
            elif op == 8:
                st = bytes(t.stream)
                out += len(st).to_bytes(4, "little") + st
        except T.TranscriptError:
            return 1000 + i, out
    return 0, out


def pack_script(ops):
    b = b""
    for op, payload in ops:
        b += bytes([op])
        if op in (2, 6):
            b += O.fe_to_bytes(payload)
        elif op in (3, 7):
            b += O.g1_to_bytes(payload)
    return b


def be_point(p):
    return p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")


def main():
    rng = random.Random(0xE7A1)
    pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(6)]
    sc = [rng.randrange(O.R) for _ in range(6)] + [0, 1, O.R - 1]
    cases = []

    def add(name, ops, proof=b""):
        rc, out = run_script(ops, proof)
        cases.append({"name": name, "script": pack_script(ops).hex(), "proof": proof.hex(), "rc": rc, "out": out.hex()})

    add("squeeze_empty", [(1, None)])
    add("double_squeeze_appends_01", [(2, sc[0]), (1, None), (1, None), (1, None)])
    add("absorb_then_squeeze", [(3, pts[0]), (2, sc[1]), (1, None), (3, pts[1]), (1, None)])
    add("absorb_32_bytes_then_squeeze_gets_01_suffix", [(2, sc[2]), (1, None)])
    proof = be_point(pts[2]) + sc[3].to_bytes(32, "big") + be_point(pts[3]) + sc[8].to_bytes(32, "big")
    add("read_points_and_scalars", [(5, None), (4, None), (1, None), (5, None), (4, None), (1, None), (1, None)], proof)
    add("write_then_finalize", [(7, pts[4]), (6, sc[4]), (1, None), (7, pts[5]), (6, sc[6]), (6, sc[7]), (1, None), (8, None)])
    add("read_scalar_non_canonical", [(4, None), (4, None)], sc[5].to_bytes(32, "big") + O.R.to_bytes(32, "big"))
    add("read_point_off_curve", [(5, None)], (pts[0][0]).to_bytes(32, "big") + ((pts[0][1] + 1) % O.P).to_bytes(32, "big"))
    add("read_point_zero_zero", [(5, None)], bytes(64))
    add("read_point_coordinate_ge_p", [(5, None)], (pts[0][0] + O.P).to_bytes(32, "big") + pts[0][1].to_bytes(32, "big"))
    add("read_past_end", [(5, None), (4, None)], be_point(pts[1]) + b"\x00" * 31)
    add("common_identity_is_error", [(2, sc[0]), (3, None), (1, None)])
    add("write_identity_is_error", [(7, None)])
    # a long absorb crossing several 136-byte rate blocks
    add("long_absorb", [(3, pts[i % 6]) for i in range(9)] + [(2, sc[i % 9]) for i in range(7)] + [(1, None), (1, None)])
    # ---- Poseidon transcript (T=5, RATE=4, R_F=8, R_P=60)
    pcases = []

    def padd(name, ops, proof=b""):
        rc, out = run_script(ops, proof, kind=1)
        pcases.append({"name": name, "script": pack_script(ops).hex(), "proof": proof.hex(), "rc": rc, "out": out.hex()})

    padd("squeeze_empty", [(1, None)])
    padd("squeeze_twice", [(1, None), (1, None)])
    padd("one_scalar", [(2, sc[0]), (1, None)])
    padd("exactly_rate_elements", [(2, sc[i]) for i in range(4)] + [(1, None)])
    padd("rate_plus_one", [(2, sc[i]) for i in range(5)] + [(1, None), (1, None)])
    padd("points_and_scalars", [(3, pts[0]), (2, sc[1]), (1, None), (3, pts[1]), (3, pts[2]), (1, None)])
    pproof = T.g1_compress(pts[2]) + sc[3].to_bytes(32, "little") + T.g1_compress(pts[3]) + sc[8].to_bytes(32, "little")
    padd("read_points_and_scalars", [(5, None), (4, None), (1, None), (5, None), (4, None), (1, None)], pproof)
    padd("write_then_finalize", [(7, pts[4]), (6, sc[4]), (1, None), (7, pts[5]), (6, sc[6]), (1, None), (8, None)])
    padd("read_scalar_non_canonical", [(4, None)], O.R.to_bytes(32, "little"))
    bad_x = next(x for x in range(2, 100) if pow((x ** 3 + 3) % O.P, (O.P - 1) // 2, O.P) != 1)
    padd("read_point_x_not_on_curve", [(5, None)], bad_x.to_bytes(32, "little"))
    padd("read_point_x_ge_p", [(5, None)], (O.P + 1).to_bytes(32, "little"))
    padd("read_identity_cannot_be_absorbed", [(5, None)], T.g1_compress(None))
    padd("read_past_end", [(4, None)], bytes(31))
    flagged = bytearray(T.g1_compress(pts[0]))
    flagged[31] |= 0x80
    padd("read_point_identity_flag_on_a_finite_point", [(5, None)], bytes(flagged))
    padd("common_identity_is_error", [(3, None)])
    rc5, mds5 = T.poseidon_spec(5, 8, 60)
    rc3, mds3 = T.poseidon_spec(3, 8, 57)
    poseidon = {
        "t5_rf8_rp60": {"rc_first": hex(rc5[0]), "rc_last": hex(rc5[-1]), "mds00": hex(mds5[0][0]), "mds44": hex(mds5[4][4]),
                        "permute_0_1_2_3_4": [hex(x) for x in T.poseidon_permute([0, 1, 2, 3, 4], 8, 60)]},
        # the public instance every Poseidon library ships (circomlib's t = 3): these three are KNOWN ANSWERS
        "t3_rf8_rp57_public": {"rc_first": hex(rc3[0]), "mds00": hex(mds3[0][0]),
                               "permute_0_1_2_word0": hex(T.poseidon_permute([0, 1, 2], 8, 57)[0])},
    }
    vectors = {"poseidon_cases": pcases, "poseidon": poseidon, "keccak256": [{"msg": m.hex(), "digest": T.keccak256(m).hex()}
                             for m in [b"", b"abc", bytes(135), bytes(136), bytes(137), bytes(range(256)) * 2]]}
    with open(os.path.join(ROOT, "tests", "golden", "evm_transcript.json"), "w") as f:
        json.dump({"cases": cases, **vectors}, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
