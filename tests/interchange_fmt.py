"""Test helper: writes a protocol (the dict form of tests/plonk_synth.py) and a whole `Snark` in the reference's OWN
serialisations, built from the struct definitions (no Rust here to produce a real dump; tools/refgen does that on a
machine with cargo):

  serde_json / bincode 1.x of `PlonkProtocol<G1Affine>`  (snark-verifier/src/verifier/plonk/protocol.rs:17-72 and the
      types it holds: util/arithmetic.rs:92-95,120-134; protocol.rs:191-196,286-330,529-547)
  `Snark { protocol, instances, proof }`                 (snark-verifier-sdk/src/lib.rs:47-53)

serde derive rules used: struct = fields in declaration order (JSON object / bincode concatenation), newtype struct
(`Rotation(i32)`) = its field, unit enum variant = "Name" / u32 index, newtype variant = {"Name": v} / index + v, tuple
variant = {"Name": [..]} / index + fields, Option = null|v / u8 tag, Vec = [..] / u64 length, usize = u64, tuples = [..].
`field_mode`: how halo2curves 0.6.0 writes a field element -- "canonical" (32-byte LE repr; hex string in JSON) or
"montgomery" (the derive on `Fr([u64; 4])`: 4 limbs of the Montgomery residue).  Both exist in released versions of
that crate; the loaders accept both."""
import json
import struct

import bn254 as O

R, P_MOD = O.R, O.P
MONT = 1 << 256


def _limbs(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


class Enc:
    def __init__(self, field_mode):
        assert field_mode in ("canonical", "montgomery")
        self.m = field_mode

    # ---- JSON
    def jfr(self, x, mod=R):
        x %= mod
        if self.m == "canonical":
            return x.to_bytes(32, "little").hex()
        return _limbs(x * MONT % mod)

    def jg1(self, pt):
        x, y = (0, 0) if pt is None else pt
        return {"x": self.jfr(x, P_MOD), "y": self.jfr(y, P_MOD)}

    # ---- bincode
    def bfr(self, x, mod=R):
        x %= mod
        if self.m == "montgomery":
            x = x * MONT % mod
        return x.to_bytes(32, "little")

    def bg1(self, pt):
        x, y = (0, 0) if pt is None else pt
        return self.bfr(x, P_MOD) + self.bfr(y, P_MOD)


def _jexpr(e, enc):
    t = e[0]
    if t == "const":
        return {"Constant": enc.jfr(e[1])}
    if t == "identity":
        return {"CommonPolynomial": "Identity"}
    if t == "lagrange":
        return {"CommonPolynomial": {"Lagrange": e[1]}}
    if t == "poly":
        return {"Polynomial": {"poly": e[1], "rotation": e[2]}}
    if t == "challenge":
        return {"Challenge": e[1]}
    if t == "neg":
        return {"Negated": _jexpr(e[1], enc)}
    if t == "sum":
        return {"Sum": [_jexpr(e[1], enc), _jexpr(e[2], enc)]}
    if t == "prod":
        return {"Product": [_jexpr(e[1], enc), _jexpr(e[2], enc)]}
    if t == "scaled":
        return {"Scaled": [_jexpr(e[1], enc), enc.jfr(e[2])]}
    if t == "dpow":
        return {"DistributePowers": [[_jexpr(x, enc) for x in e[1]], _jexpr(e[2], enc)]}
    raise ValueError(t)


def protocol_to_json_obj(pr, field_mode="canonical"):
    enc = Enc(field_mode)
    d = pr["domain"]
    ick = pr.get("instance_committing_key")
    return {
        "domain": {"k": d.k, "n": d.n, "n_inv": enc.jfr(d.n_inv), "gen": enc.jfr(d.gen), "gen_inv": enc.jfr(d.gen_inv)},
        "preprocessed": [enc.jg1(p) for p in pr["preprocessed"]],
        "num_instance": list(pr["num_instance"]),
        "num_witness": list(pr["num_witness"]),
        "num_challenge": list(pr["num_challenge"]),
        "evaluations": [{"poly": p, "rotation": r} for p, r in pr["evaluations"]],
        "queries": [{"poly": p, "rotation": r} for p, r in pr["queries"]],
        "quotient": {"chunk_degree": pr["quotient"]["chunk_degree"], "num_chunk": pr["quotient"]["num_chunk"],
                     "numerator": _jexpr(pr["quotient"]["numerator"], enc)},
        "transcript_initial_state": None if pr.get("transcript_initial_state") is None else enc.jfr(pr["transcript_initial_state"]),
        "instance_committing_key": None if ick is None else {
            "bases": [enc.jg1(p) for p in ick["bases"]],
            "constant": None if ick.get("constant") is None else enc.jg1(ick["constant"])},
        "linearization": pr.get("linearization"),
        "accumulator_indices": [[[i, j] for i, j in idx] for idx in pr["accumulator_indices"]],
    }


def protocol_to_json(pr, field_mode="canonical", indent=None):
    return json.dumps(protocol_to_json_obj(pr, field_mode), indent=indent).encode()


def snark_to_json(pr, instances, proof, field_mode="canonical"):
    enc = Enc(field_mode)
    return json.dumps({"protocol": protocol_to_json_obj(pr, field_mode),
                       "instances": [[enc.jfr(x) for x in col] for col in instances],
                       "proof": list(proof)}).encode()


_u64 = lambda x: struct.pack("<Q", x)
_u32 = lambda x: struct.pack("<I", x)
_i32 = lambda x: struct.pack("<i", x)


def _bexpr(e, enc):
    t = e[0]
    if t == "const":
        return _u32(0) + enc.bfr(e[1])
    if t == "identity":
        return _u32(1) + _u32(0)
    if t == "lagrange":
        return _u32(1) + _u32(1) + _i32(e[1])
    if t == "poly":
        return _u32(2) + _u64(e[1]) + _i32(e[2])
    if t == "challenge":
        return _u32(3) + _u64(e[1])
    if t == "neg":
        return _u32(4) + _bexpr(e[1], enc)
    if t == "sum":
        return _u32(5) + _bexpr(e[1], enc) + _bexpr(e[2], enc)
    if t == "prod":
        return _u32(6) + _bexpr(e[1], enc) + _bexpr(e[2], enc)
    if t == "scaled":
        return _u32(7) + _bexpr(e[1], enc) + enc.bfr(e[2])
    if t == "dpow":
        return _u32(8) + _u64(len(e[1])) + b"".join(_bexpr(x, enc) for x in e[1]) + _bexpr(e[2], enc)
    raise ValueError(t)


def protocol_to_bincode(pr, field_mode="canonical"):
    enc = Enc(field_mode)
    d = pr["domain"]
    out = _u64(d.k) + _u64(d.n) + enc.bfr(d.n_inv) + enc.bfr(d.gen) + enc.bfr(d.gen_inv)
    out += _u64(len(pr["preprocessed"])) + b"".join(enc.bg1(p) for p in pr["preprocessed"])
    for key in ("num_instance", "num_witness", "num_challenge"):
        out += _u64(len(pr[key])) + b"".join(_u64(x) for x in pr[key])
    for key in ("evaluations", "queries"):
        out += _u64(len(pr[key])) + b"".join(_u64(p) + _i32(r) for p, r in pr[key])
    q = pr["quotient"]
    out += _u64(q["chunk_degree"]) + _u64(q["num_chunk"]) + _bexpr(q["numerator"], enc)
    tis = pr.get("transcript_initial_state")
    out += b"\x00" if tis is None else b"\x01" + enc.bfr(tis)
    ick = pr.get("instance_committing_key")
    if ick is None:
        out += b"\x00"
    else:
        out += b"\x01" + _u64(len(ick["bases"])) + b"".join(enc.bg1(p) for p in ick["bases"])
        out += b"\x00" if ick.get("constant") is None else b"\x01" + enc.bg1(ick["constant"])
    lin = pr.get("linearization")
    out += b"\x00" if lin is None else b"\x01" + _u32({"WithoutConstant": 0, "MinusVanishingTimesQuotient": 1}[lin])
    out += _u64(len(pr["accumulator_indices"]))
    for idx in pr["accumulator_indices"]:
        out += _u64(len(idx)) + b"".join(_u64(i) + _u64(j) for i, j in idx)
    return out


def snark_to_bincode(pr, instances, proof, field_mode="canonical"):
    enc = Enc(field_mode)
    out = protocol_to_bincode(pr, field_mode)
    out += _u64(len(instances)) + b"".join(_u64(len(c)) + b"".join(enc.bfr(x) for x in c) for c in instances)
    return out + _u64(len(proof)) + bytes(proof)
