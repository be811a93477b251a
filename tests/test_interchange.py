"""N3 interchange: the reference's own serialisations of `PlonkProtocol` (serde_json, bincode;
snark-verifier/src/verifier/plonk/protocol.rs:17-72) and of the SDK's `Snark` (bincode written by `gen_snark`,
snark-verifier-sdk/src/halo2.rs:266-281; lib.rs:47-53) are read by the host mirror (host/serde_json.hpp behind
`snarkv_host_protocol_parse` / `snarkv_host_snark_parse`) into the same protocol the packed wire form gives.

No real dump exists in this container (tools/refgen needs cargo), so the inputs are written from the struct
definitions by tests/interchange_fmt.py, in BOTH field encodings a halo2curves release may use.  When
tests/golden/ref_snark.{json,bin} (the output of tools/refgen) are present they are loaded too."""
import json
import os
import random

import pytest

import bn254 as O
import interchange_fmt as X
import plonk_synth as S
from snark_verifier_amd import host_api as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _protocols():
    rng = random.Random(0x1237)
    out = [S.standard_plonk_protocol(rng)[0],
           S.standard_plonk_protocol(rng, linearization="WithoutConstant", accumulator_rows=[range(0, 16)], num_instance=(16,))[0],
           S.standard_plonk_protocol(rng, committed_instances=True, initial_state=False)[0]]
    for seed in range(6):
        r = random.Random(900 + seed)
        out.append(S.random_protocol(r, r.choice([None, "WithoutConstant", "MinusVanishingTimesQuotient"]))[0])
    return out


@pytest.mark.parametrize("field_mode", ["canonical", "montgomery"])
def test_protocol_serde_json_and_bincode_equal_packed_form(field_mode):
    for pr in _protocols():
        packed = S.pack_protocol(pr)
        assert H.Protocol(packed).pack() == packed  # parser and writer of the packed form are inverse
        for indent in (None, 2):
            j = H.Protocol(X.protocol_to_json(pr, field_mode, indent), H.PROTOCOL_SERDE_JSON)
            assert j.pack() == packed
        b = H.Protocol(X.protocol_to_bincode(pr, field_mode), H.PROTOCOL_BINCODE)
        assert b.pack() == packed


@pytest.mark.parametrize("field_mode", ["canonical", "montgomery"])
def test_snark_bincode_and_json(field_mode):
    rng = random.Random(77)
    pr = _protocols()[0]
    inst = [[rng.randrange(O.R) for _ in range(m)] for m in pr["num_instance"]]
    proof = bytes(rng.randrange(256) for _ in range(1440))
    for fmt, data in ((H.PROTOCOL_BINCODE, X.snark_to_bincode(pr, inst, proof, field_mode)),
                      (H.PROTOCOL_SERDE_JSON, X.snark_to_json(pr, inst, proof, field_mode))):
        s = H.Snark(data, fmt)
        assert s.protocol.pack() == S.pack_protocol(pr)
        assert s.instances == S.pack_instances(inst)
        assert s.proof == proof
        s.close()


def test_malformed_inputs_are_errors_not_crashes():
    pr = _protocols()[0]
    good_b = X.protocol_to_bincode(pr)
    for cut in (0, 7, 50, len(good_b) // 2, len(good_b) - 1):
        with pytest.raises(H.HostError):
            H.Protocol(good_b[:cut], H.PROTOCOL_BINCODE)
    with pytest.raises(H.HostError):
        H.Protocol(good_b + b"\x00", H.PROTOCOL_BINCODE)  # trailing bytes
    bad = bytearray(good_b)
    bad[16] ^= 1  # n_inv no longer inverts n under either encoding
    with pytest.raises(H.HostError):
        H.Protocol(bytes(bad), H.PROTOCOL_BINCODE)
    obj = X.protocol_to_json_obj(pr)
    for mutate in (lambda o: o.pop("queries"),
                   lambda o: o["quotient"].__setitem__("numerator", {"Bogus": 1}),
                   lambda o: o.__setitem__("linearization", "Sideways"),
                   lambda o: o["domain"].__setitem__("n", 5),
                   lambda o: o["preprocessed"].__setitem__(0, {"x": "00", "y": "00"})):
        o = json.loads(json.dumps(obj))
        mutate(o)
        with pytest.raises(H.HostError):
            H.Protocol(json.dumps(o).encode(), H.PROTOCOL_SERDE_JSON)
    for text in (b"", b"[", b'{"domain": }', b'{"a": "\\u12"}', b"nul"):
        with pytest.raises(H.HostError):
            H.Protocol(text, H.PROTOCOL_SERDE_JSON)


def _mutations(rng, data, count):
    """byte flips, 0xFF / 0x00 overwrites of a length-sized window, truncations, splices"""
    for _ in range(count):
        b = bytearray(data)
        kind = rng.randrange(5)
        if kind == 0:
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            at = rng.randrange(len(b))
            b[at:at + 8] = b"\xff" * min(8, len(b) - at)
        elif kind == 2:
            at = rng.randrange(len(b))
            b[at:at + 8] = b"\x00" * min(8, len(b) - at)
        elif kind == 3:
            b = b[:rng.randrange(len(b))]
        else:
            at, ln = rng.randrange(len(b)), rng.randrange(1, 64)
            b[at:at + ln] = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 64)))
        yield bytes(b)


def _as_bytes(x):
    return x.encode() if isinstance(x, str) else bytes(x)


def test_mutated_inputs_never_crash_the_readers():
    """Seeded mutation fuzz of the four readers (protocol / snark x bincode / JSON): every input either parses (and
    then packs without error) or is refused with an error code -- no crash, no hang, no runaway allocation from a
    corrupted length prefix."""
    rng = random.Random(0xF022)
    pr = _protocols()[0]
    inst = [[rng.randrange(O.R) for _ in range(m)] for m in pr["num_instance"]]
    proof = bytes(rng.randrange(256) for _ in range(320))
    seeds = ((H.Protocol, H.PROTOCOL_BINCODE, X.protocol_to_bincode(pr)),
             (H.Protocol, H.PROTOCOL_SERDE_JSON, _as_bytes(X.protocol_to_json(pr))),
             (H.Snark, H.PROTOCOL_BINCODE, X.snark_to_bincode(pr, inst, proof)),
             (H.Snark, H.PROTOCOL_SERDE_JSON, _as_bytes(X.snark_to_json(pr, inst, proof))))
    parsed = refused = 0
    for cls, fmt, data in seeds:
        for bad in _mutations(rng, data, 250):
            try:
                obj = cls(bad, fmt)
            except H.HostError:
                refused += 1
                continue
            parsed += 1
            (obj.protocol if cls is H.Snark else obj).pack()
            obj.close()
    assert refused > 500 and parsed + refused == 1000


def test_reference_generated_snark_if_present():
    """tools/refgen (Rust; needs cargo + network, neither here) writes ref_snark.bin / ref_snark.json with the
    reference's own serde: when a maintainer has dropped them into tests/golden/, both must load and agree."""
    pb, pj = (os.path.join(ROOT, "tests", "golden", n) for n in ("ref_snark.bin", "ref_snark.json"))
    if not (os.path.exists(pb) and os.path.exists(pj)):
        pytest.skip("NO REFERENCE-GENERATED FIXTURE: tests/golden/ref_snark.{bin,json} absent (run tools/refgen on a machine "
                    "with Rust; parity of the interchange formats stays UNPINNED until then)")
    a, b = H.Snark(open(pb, "rb").read(), H.PROTOCOL_BINCODE), H.Snark(open(pj, "rb").read(), H.PROTOCOL_SERDE_JSON)
    assert a.protocol.pack() == b.protocol.pack() and a.instances == b.instances and a.proof == b.proof
