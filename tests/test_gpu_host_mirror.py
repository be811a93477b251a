"""GPU parity of the C++ host mirror (snark-verifier_amd/host/) -- `Msm`,
`KzgAs<Gwc19|Bdfg21>`, the decider, `LimbsEncoding` -- with every MSM / pairing
on the MI355X, against oracle/kzg.py.  Mirrors how the reference's callers use
the path (examples/evm-verifier-with-accumulator.rs:357-385)."""
import ctypes
import random

import pytest

import bn254 as O
import coracle as C
import kzg as K
from hostfmt import fr, g1, load_host_lib, msm_expected, pack_bdfg21, pack_commitments, pack_gwc19

pytestmark = pytest.mark.gpu

SECRET = 0x1F2E3D4C5B6A79887766554433221100AABBCCDDEEFF


@pytest.fixture(scope="module")
def H():
    return load_host_lib()


def _buf(n):
    return ctypes.create_string_buffer(n)


def test_msm_evaluate_constant_first_and_dedup(H):
    rng = random.Random(1)
    pts = [O.g1_from_bytes(C.sample_points(9, 1, first=i)) for i in range(5)]
    m = K.Msm.const(rng.randrange(O.R))
    for p in pts + [pts[1], pts[3]]:  # duplicates are merged by `push` (msm.rs:109-116)
        m = m + K.Msm.base(p) * rng.randrange(O.R)
    assert len(m.bases) == 5
    o = _buf(64)
    assert H.hd_msm_evaluate(g1(O.G1_GEN) + pack_commitments([m]), 1, o) == 0
    assert o.raw == msm_expected(m, O.G1_GEN)
    # constant without generator: the reference panics (msm.rs:93)
    assert H.hd_msm_evaluate(g1(O.G1_GEN) + pack_commitments([m]), 0, o) == -100
    m2 = K.Msm.base(pts[0]) * 5 - K.Msm.base(pts[0]) * 5  # scalar 0 stays a term
    assert H.hd_msm_evaluate(g1(O.G1_GEN) + pack_commitments([m2]), 0, o) == 0 and o.raw == b"\x00" * 64


def test_gwc19_verify_matches_oracle_and_is_valid(H):
    rng = random.Random(2)
    inst = K.synth_gwc19_instance(rng, SECRET)
    lhs, rhs = K.gwc19_msms(inst["g"], inst["commitments"], inst["z"], inst["queries"], inst["v"], inst["ws"], inst["u"])
    o, sizes = _buf(128), (ctypes.c_uint32 * 2)()
    assert H.hd_gwc19_verify(pack_gwc19(inst), o, sizes) == 0
    assert list(sizes) == [21, 3]
    assert o.raw == msm_expected(lhs, inst["g"]) + msm_expected(rhs, inst["g"])
    assert O.g1_from_bytes(o.raw[:64]) == O.g1_mul(O.g1_from_bytes(o.raw[64:]), SECRET)


def test_bdfg21_verify_matches_oracle_and_is_valid(H):
    rng = random.Random(3)
    inst = K.synth_bdfg21_instance(rng, SECRET)
    lhs, rhs = K.bdfg21_msms(inst["g"], inst["commitments"], inst["z"], inst["queries"], inst["mu"], inst["gamma"],
                             inst["w"], inst["z_prime"], inst["w_prime"])
    o, sizes = _buf(128), (ctypes.c_uint32 * 2)()
    assert H.hd_bdfg21_verify(pack_bdfg21(inst), o, sizes) == 0
    assert list(sizes) == [20, 1]
    assert o.raw == msm_expected(lhs, inst["g"]) + msm_expected(rhs, inst["g"])
    assert O.g1_from_bytes(o.raw[:64]) == O.g1_mul(O.g1_from_bytes(o.raw[64:]), SECRET)


def _mock_accumulators(rng, m):
    """valid accumulators (s*a*G, a*G): the reference's mock fixture idea
    (system/halo2/test/kzg.rs:30-46)"""
    out = []
    for _ in range(m):
        a = rng.randrange(1, O.R)
        rhs = C.g1_mul(g1(O.G1_GEN), fr(a))
        out.append(C.g1_mul(rhs, fr(SECRET)) + rhs)
    return out


def test_kzg_as_verify_and_create_proof(H):
    rng = random.Random(4)
    accs = _mock_accumulators(rng, 10)
    r = rng.randrange(O.R)
    o = _buf(128)
    assert H.hd_kzg_as_verify(b"".join(accs), 10, fr(r), None, o) == 0
    pairs = [(O.g1_from_bytes(a[:64]), O.g1_from_bytes(a[64:])) for a in accs]
    exp = K.kzg_as_verify(pairs, r)
    assert o.raw == g1(exp[0]) + g1(exp[1])
    # zk: blind pair (s*b*G, b*G) read from the transcript goes last (accumulation.rs:46-50)
    b = rng.randrange(O.R)
    blind = (O.g1_mul(O.G1_GEN, SECRET * b % O.R), O.g1_mul(O.G1_GEN, b))
    assert H.hd_kzg_as_verify(b"".join(accs), 10, fr(r), g1(blind[0]) + g1(blind[1]), o) == 0
    exp = K.kzg_as_verify(pairs, r, blind)
    assert o.raw == g1(exp[0]) + g1(exp[1])
    # prover twin writes that same blind pair and lands on the same accumulator
    pk = g1(O.G1_GEN) + g1(O.g1_mul(O.G1_GEN, SECRET))
    o2, wr = _buf(128), _buf(128)
    assert H.hd_kzg_as_create_proof(b"".join(accs), 10, fr(r), pk, fr(b), o2, wr) == 0
    assert wr.raw == g1(blind[0]) + g1(blind[1]) and o2.raw == o.raw
    # no instances: the reference asserts (accumulation.rs:122)
    assert H.hd_kzg_as_verify(b"", 0, fr(r), None, o) == -100


def test_decide_and_decide_all(H):
    rng = random.Random(5)
    g2 = O.g2_to_bytes(O.G2_GEN)
    s_g2 = O.g2_to_bytes(O.g2_mul(O.G2_GEN, SECRET))
    accs = _mock_accumulators(rng, 6)
    for one_by_one in (0, 1):
        assert H.hd_decide_all(g1(O.G1_GEN), g2, s_g2, b"".join(accs), 6, one_by_one) == 1
    bad = list(accs)
    bad[4] = C.g1_mul(g1(O.G1_GEN), fr(12345)) + accs[4][64:]
    for one_by_one in (0, 1):
        assert H.hd_decide_all(g1(O.G1_GEN), g2, s_g2, b"".join(bad), 6, one_by_one) == 0
    assert H.hd_decide_all(g1(O.G1_GEN), g2, s_g2, b"", 0, 0) == 1  # decide_all(vec![]) == Ok(())


def test_limbs_encoding_roundtrip_and_panics(H):
    rng = random.Random(6)
    acc = _mock_accumulators(rng, 1)[0]
    pair = (O.g1_from_bytes(acc[:64]), O.g1_from_bytes(acc[64:]))
    limbs = K.accumulator_to_limbs(pair)
    o, lo = _buf(128), _buf(16 * 32)
    assert H.hd_limbs_roundtrip(b"".join(fr(x) for x in limbs), o, lo) == 0
    assert o.raw == acc
    assert lo.raw == b"".join(fr(x) for x in limbs)
    off = list(limbs)
    off[4] ^= 1  # y of lhs perturbed -> off-curve: `from_xy().unwrap()` panics
    assert H.hd_limbs_roundtrip(b"".join(fr(x) for x in off), o, lo) == -100


def test_config3_accumulate_64_standard_plonk_proofs(H):
    """BASELINE config 3: KzgAs<Gwc19> accumulation of 64 StandardPlonk-shaped
    proofs, bit-exact accumulator vs the CPU oracle, then the decider accepts."""
    rng = random.Random(7)
    accs, exp_accs = [], []
    o, sizes = _buf(128), (ctypes.c_uint32 * 2)()
    for _ in range(64):
        inst = K.synth_gwc19_instance(rng, SECRET)
        assert H.hd_gwc19_verify(pack_gwc19(inst), o, sizes) == 0
        lhs, rhs = K.gwc19_msms(inst["g"], inst["commitments"], inst["z"], inst["queries"], inst["v"], inst["ws"], inst["u"])
        exp = msm_expected(lhs, inst["g"]) + msm_expected(rhs, inst["g"])
        assert o.raw == exp
        accs.append(o.raw)
    r = rng.randrange(O.R)
    assert H.hd_kzg_as_verify(b"".join(accs), 64, fr(r), None, o) == 0
    pw = K.powers(r, 64)
    s = b"".join(fr(p) for p in pw)
    exp = C.msm_naive(s, b"".join(a[:64] for a in accs)) + C.msm_naive(s, b"".join(a[64:] for a in accs))
    assert o.raw == exp
    g2 = O.g2_to_bytes(O.G2_GEN)
    s_g2 = O.g2_to_bytes(O.g2_mul(O.G2_GEN, SECRET))
    assert H.hd_decide_all(g1(O.G1_GEN), g2, s_g2, o.raw, 1, 0) == 1


def test_golden_kzg_layer_fixture(H):
    """Committed fixture tests/golden/kzg_layer.json: the C++ mirror + HIP kernels
    reproduce every accumulator byte for byte."""
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", "kzg_layer.json")))
    o, sizes = _buf(128), (ctypes.c_uint32 * 2)()
    for case in g["gwc19"]:
        assert H.hd_gwc19_verify(bytes.fromhex(case["input"]), o, sizes) == 0
        assert o.raw.hex() == case["accumulator"] and list(sizes) == case["msm_sizes"]
    for case in g["bdfg21"]:
        assert H.hd_bdfg21_verify(bytes.fromhex(case["input"]), o, sizes) == 0
        assert o.raw.hex() == case["accumulator"] and list(sizes) == case["msm_sizes"]
    ka = g["kzg_as"]
    accs = bytes.fromhex(ka["accumulators"])
    m = len(accs) // 128
    assert H.hd_kzg_as_verify(accs, m, bytes.fromhex(ka["r"]), None, o) == 0 and o.raw.hex() == ka["result"]
    assert H.hd_kzg_as_verify(accs, m, bytes.fromhex(ka["r"]), bytes.fromhex(ka["blind"]), o) == 0
    assert o.raw.hex() == ka["result_zk"]
    assert H.hd_decide_all(g1(O.G1_GEN), bytes.fromhex(g["g2"]), bytes.fromhex(g["s_g2"]), bytes.fromhex(ka["result"]), 1, 0) == 1
    lo = _buf(16 * 32)
    assert H.hd_limbs_roundtrip(bytes.fromhex(g["limbs"]["limbs"]), o, lo) == 0
    assert o.raw.hex() == g["limbs"]["accumulator"] and lo.raw.hex() == g["limbs"]["limbs"]


@pytest.mark.parametrize("kind", ["evm", "poseidon"])
def test_kzg_as_over_the_real_transcripts(H, kind):
    """KzgAs prover -> proof bytes -> verifier with the REAL transcripts
    (host/transcript.hpp; reference system/halo2/transcript/evm.rs and halo2.rs):
    the challenge r is derived, not supplied; both sides and the oracle must land
    on the same accumulator, and it must still satisfy the pairing check."""
    import transcript as T

    rng = random.Random(40)
    fn = H.hd_kzg_as_evm_roundtrip if kind == "evm" else H.hd_kzg_as_poseidon_roundtrip
    fn.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p,
                   ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    for m, zk in ((1, False), (7, False), (7, True), (64, True) if kind == "evm" else (9, True)):
        accs = _mock_accumulators(rng, m)
        pairs = [(O.g1_from_bytes(a[:64]), O.g1_from_bytes(a[64:])) for a in accs]
        b = rng.randrange(1, O.R)
        pk = g1(O.G1_GEN) + g1(O.g1_mul(O.G1_GEN, SECRET)) if zk else None
        out = _buf(4096)
        n = ctypes.c_size_t(0)
        assert fn(b"".join(accs), m, pk, fr(b), out, len(out), ctypes.byref(n)) == 0
        raw = out.raw[:n.value]
        plen = int.from_bytes(raw[:4], "little")
        proof, acc_p, acc_v, r_got = raw[4:4 + plen], raw[4 + plen:132 + plen], raw[132 + plen:260 + plen], raw[260 + plen:]
        # oracle: same transcript, same algebra
        t = T.EvmTranscript() if kind == "evm" else T.PoseidonTranscript()
        for lhs, rhs in pairs:
            t.common_ec_point(lhs)
            t.common_ec_point(rhs)
        blind = None
        if zk:
            blind = (O.g1_mul(O.G1_GEN, SECRET * b % O.R), O.g1_mul(O.G1_GEN, b))
            t.write_ec_point(blind[0])
            t.write_ec_point(blind[1])
        r = t.squeeze_challenge()
        exp = K.kzg_as_verify(pairs, r, blind)
        assert proof == t.finalize() and len(proof) == ((128 if kind == "evm" else 64) if zk else 0)
        assert r_got == fr(r)
        assert acc_p == acc_v == g1(exp[0]) + g1(exp[1])
        g2 = O.g2_to_bytes(O.G2_GEN)
        s_g2 = O.g2_to_bytes(O.g2_mul(O.G2_GEN, SECRET))
        assert H.hd_decide_all(g1(O.G1_GEN), g2, s_g2, acc_v, 1, 1) == 1


@pytest.mark.parametrize("kind", ["evm", "poseidon"])
def test_product_c_api_kzg_as_create_proof_and_verify(kind):
    """The PRODUCT C API (include/snarkv_host.h, no test hooks): `KzgAs::create_proof` in full -- zk blind branch
    (accumulation.rs:163-175) and both transcripts -- and `KzgAsProof::read` + `KzgAs::verify` on caller-supplied proof
    bytes (accumulation.rs:114-137, 41-63), against the oracle's transcript and algebra; a tampered or truncated proof
    and trailing bytes are errors, not accumulators."""
    import transcript as T
    from snark_verifier_amd import host_api as HA

    tk = HA.TRANSCRIPT_EVM if kind == "evm" else HA.TRANSCRIPT_POSEIDON
    rng = random.Random(41)
    dk = HA.DecidingKey(g1(O.G1_GEN) + O.g2_to_bytes(O.G2_GEN) + O.g2_to_bytes(O.g2_mul(O.G2_GEN, SECRET)))
    for m, zk in ((1, False), (5, False), (5, True), (33, True)):
        accs = _mock_accumulators(rng, m)
        pairs = [(O.g1_from_bytes(a[:64]), O.g1_from_bytes(a[64:])) for a in accs]
        b = rng.randrange(1, O.R)
        pk = g1(O.G1_GEN) + g1(O.g1_mul(O.G1_GEN, SECRET)) if zk else None
        acc, proof, r = HA.kzg_as_create_proof(b"".join(accs), tk, pk, fr(b) if zk else None)
        t = T.EvmTranscript() if kind == "evm" else T.PoseidonTranscript()
        for lhs, rhs in pairs:
            t.common_ec_point(lhs)
            t.common_ec_point(rhs)
        blind = None
        if zk:
            blind = (O.g1_mul(O.G1_GEN, SECRET * b % O.R), O.g1_mul(O.G1_GEN, b))
            t.write_ec_point(blind[0])
            t.write_ec_point(blind[1])
        r_exp = t.squeeze_challenge()
        exp = K.kzg_as_verify(pairs, r_exp, blind)
        assert proof == t.finalize() and len(proof) == ((128 if kind == "evm" else 64) if zk else 0)
        assert r == fr(r_exp) and acc == g1(exp[0]) + g1(exp[1])
        # the verifier's side on those bytes
        acc_v, r_v = HA.kzg_as_verify(b"".join(accs), proof, tk, zk)
        assert acc_v == acc and r_v == r
        assert HA.kzg_decide(dk, acc_v) is True
        if not zk:
            assert HA.kzg_as_accumulate(b"".join(accs))[0] == acc or kind != "evm"  # the round-2 entry point = this one, non-zk / EVM
        with pytest.raises(HA.HostError):  # bytes left over
            HA.kzg_as_verify(b"".join(accs), proof + b"\x00" * 32, tk, zk)
        if zk:
            with pytest.raises(HA.HostError):  # truncated: Error::Transcript
                HA.kzg_as_verify(b"".join(accs), proof[:-1], tk, zk)
            bad = bytearray(proof)
            bad[5] ^= 1  # no longer a curve point (EVM: x | y big-endian; Poseidon: compressed x) -- or another point: a different accumulator
            try:
                acc_b, _ = HA.kzg_as_verify(b"".join(accs), bytes(bad), tk, zk)
                assert acc_b != acc
            except HA.HostError:
                pass
    dk.close()
