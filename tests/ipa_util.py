"""Shared helpers of the IPA tests: fixture loading and byte packing for the host driver."""
import json
import os
import struct

import bn254 as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_cases(section="cases"):
    with open(os.path.join(ROOT, "tests", "golden", "ipa.json")) as f:
        return json.load(f)[section]


def bgh19_case(c):
    """-> (g, h, s, commitment points, x, queries [(poly, shift, eval)], proof bytes, accumulator)"""
    g, h, s = case_key(c)
    coms = [O.g1_from_bytes(bytes.fromhex(v)) for v in c["commitments"]]
    queries = [(p, O.fe_from_bytes(bytes.fromhex(sh)), O.fe_from_bytes(bytes.fromhex(ev))) for p, sh, ev in c["queries"]]
    return g, h, s, coms, O.fe_from_bytes(bytes.fromhex(c["x"])), queries, bytes.fromhex(c["proof"]), acc_from_json(c["accumulator"])


def case_key(c):
    g = [O.g1_from_bytes(bytes.fromhex(x)) for x in c["g"]]
    h = O.g1_from_bytes(bytes.fromhex(c["h"]))
    s = O.g1_from_bytes(bytes.fromhex(c["s"])) if c["s"] else None
    return g, h, s


def acc_from_json(a):
    return [O.fe_from_bytes(bytes.fromhex(x)) for x in a["xi"]], O.g1_from_bytes(bytes.fromhex(a["u"]))


def pack_acc(acc):
    return b"".join(O.fe_to_bytes(x) for x in acc[0]) + O.g1_to_bytes(acc[1])


def pack_svk(k, g0, h, s):
    return struct.pack("<II", k, 1 if s is not None else 0) + O.g1_to_bytes(g0) + O.g1_to_bytes(h) + (
        O.g1_to_bytes(s) if s is not None else b"")
