"""CPU: the Keccak/EVM transcript of the host mirror (snark-verifier_amd/host/transcript.hpp)
vs the oracle restatement (oracle/transcript.py) of
snark-verifier/src/system/halo2/transcript/evm.rs:175-268,373-398, and both vs the
committed fixture tests/golden/evm_transcript.json."""
import ctypes
import hashlib
import json
import os
import random

import pytest

import bn254 as O
import transcript as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "evm_transcript.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def H():
    from hostfmt import load_host_lib

    L = load_host_lib()
    L.hd_keccak256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    L.hd_keccak256.restype = None
    L.hd_evm_transcript_script.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                           ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    return L


def run_cpp(H, script, proof):
    out = ctypes.create_string_buffer(1 << 16)
    n = ctypes.c_size_t(0)
    rc = H.hd_evm_transcript_script(script, len(script), proof, len(proof), out, len(out), ctypes.byref(n))
    return rc, out.raw[:n.value]


def test_oracle_keccak_is_pinned(golden):
    # the permutation + sponge reproduce hashlib's SHA3-256 (same permutation, 0x06 padding) ...
    rng = random.Random(1)
    for n in [0, 1, 31, 32, 64, 135, 136, 137, 271, 272, 273, 1000]:
        m = bytes(rng.randrange(256) for _ in range(n))
        assert T.sha3_256(m) == hashlib.sha3_256(m).digest()
    # ... and the public Keccak-256 (0x01 padding, the EVM's) known answers
    assert T.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert T.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    for v in golden["keccak256"]:
        assert T.keccak256(bytes.fromhex(v["msg"])).hex() == v["digest"]


def test_cpp_keccak_matches(H, golden):
    o = ctypes.create_string_buffer(32)
    for v in golden["keccak256"]:
        m = bytes.fromhex(v["msg"])
        H.hd_keccak256(m, len(m), o)
        assert o.raw.hex() == v["digest"]
    rng = random.Random(2)
    for n in list(range(0, 140)) + [271, 272, 273, 407, 408, 409, 5000]:
        m = bytes(rng.randrange(256) for _ in range(n))
        H.hd_keccak256(m, n, o)
        assert o.raw == T.keccak256(m), n


def test_golden_cases_cpp_and_oracle(H, golden):
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_t", os.path.join(ROOT, "tests", "golden", "gen_golden_transcript.py"))
    for case in golden["cases"]:
        script, proof = bytes.fromhex(case["script"]), bytes.fromhex(case["proof"])
        rc, out = run_cpp(H, script, proof)
        assert rc == case["rc"], case["name"]
        assert out.hex() == case["out"], case["name"]
    assert spec is not None  # the generator that made the fixture is committed next to it


def _random_script(rng, pts):
    ops, proof = [], b""
    for _ in range(rng.randrange(1, 25)):
        op = rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8])
        if op in (2, 6):
            ops.append((op, rng.choice([0, 1, O.R - 1, rng.randrange(O.R)])))
        elif op in (3, 7):
            ops.append((op, rng.choice(pts)))
        else:
            ops.append((op, None))
    # a proof stream with mostly valid items, sometimes a bad one
    for op, _ in ops:
        if op == 4:
            v = rng.randrange(O.R) if rng.random() < 0.9 else O.R + rng.randrange(1000)
            proof += v.to_bytes(32, "big")
        elif op == 5:
            p = rng.choice(pts[:-1])
            if rng.random() < 0.1:
                p = (p[0], (p[1] + 1) % O.P)
            proof += p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")
    if rng.random() < 0.1:
        proof = proof[:max(0, len(proof) - 7)]
    return ops, proof


def test_random_scripts_cpp_vs_oracle(H):
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_t", os.path.join(ROOT, "tests", "golden", "gen_golden_transcript.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rng = random.Random(77)
    pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(5)] + [None]
    for _ in range(150):
        ops, proof = _random_script(rng, pts)
        exp_rc, exp_out = gen.run_script(ops, proof)
        rc, out = run_cpp(H, gen.pack_script(ops), proof)
        assert (rc, out) == (exp_rc, exp_out)


def test_challenge_is_hash_mod_r():
    """u256_to_fe: the challenge is the big-endian hash reduced mod r (hash space is ~5.3 r)."""
    t = T.EvmTranscript()
    t.common_scalar(5)
    c = t.squeeze_challenge()
    h = T.keccak256((5).to_bytes(32, "big") + b"\x01")  # 32 buffered bytes -> 0x01 suffix (evm.rs:188-193)
    assert c == int.from_bytes(h, "big") % O.R
    assert bytes(t.buf) == h


# ---------------------------------------------------------------- Poseidon
def _setup_poseidon(H):
    H.hd_transcript_script.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                       ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    H.hd_poseidon_spec.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_size_t)]
    H.hd_poseidon_permute.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]


def run_cpp_kind(H, kind, script, proof):
    out = ctypes.create_string_buffer(1 << 16)
    n = ctypes.c_size_t(0)
    rc = H.hd_transcript_script(kind, script, len(script), proof, len(proof), out, len(out), ctypes.byref(n))
    return rc, out.raw[:n.value]


def test_poseidon_oracle_is_pinned_by_the_public_instance(golden):
    """Grain LFSR + Cauchy MDS + permutation reproduce the PUBLIC BN254 t=3 instance
    (the one circomlib & co. ship): first round constant, MDS[0][0] and the known
    answer poseidon([1, 2]).  These three hex strings are public constants."""
    rc, mds = T.poseidon_spec(3, 8, 57)
    assert rc[0] == 0x0EE9A592BA9A9518D05986D656F40C2114C4993C11BB29938D21D47304CD8E6E
    assert mds[0][0] == 0x109B7F411BA0E4C9B2B70CAF5C36A7B194BE7C11AD24378BFEDB68592BA8118B
    assert T.poseidon_permute([0, 1, 2], 8, 57)[0] == 0x115CC0F5E7D690413DF64C6B9662E9CF2A3617F2743245519E19607A4417189A
    g = golden["poseidon"]["t3_rf8_rp57_public"]
    assert int(g["rc_first"], 16) == rc[0] and int(g["mds00"], 16) == mds[0][0]


def test_cpp_poseidon_spec_and_permutation(H, golden):
    _setup_poseidon(H)
    for (t, rf, rp) in ((3, 8, 57), (5, 8, 60)):
        rc, mds = T.poseidon_spec(t, rf, rp)
        out = ctypes.create_string_buffer(32 * ((rf + rp) * t + t * t))
        n = ctypes.c_size_t(0)
        assert H.hd_poseidon_spec(t, rf, rp, out, len(out), ctypes.byref(n)) == 0
        vals = [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(n.value // 32)]
        assert vals[:len(rc)] == rc
        assert vals[len(rc):] == [mds[i][j] for i in range(t) for j in range(t)]
        rng = random.Random(t)
        for _ in range(3):
            st = [rng.randrange(O.R) for _ in range(t)]
            buf = ctypes.create_string_buffer(b"".join(O.fe_to_bytes(x) for x in st), 32 * t)
            assert H.hd_poseidon_permute(t, rf, rp, buf) == 0
            got = [int.from_bytes(buf.raw[32 * i:32 * i + 32], "little") for i in range(t)]
            assert got == T.poseidon_permute(st, rf, rp)
    g = golden["poseidon"]["t5_rf8_rp60"]
    buf = ctypes.create_string_buffer(b"".join(O.fe_to_bytes(x) for x in range(5)), 160)
    assert H.hd_poseidon_permute(5, 8, 60, buf) == 0
    assert [hex(int.from_bytes(buf.raw[32 * i:32 * i + 32], "little")) for i in range(5)] == g["permute_0_1_2_3_4"]


def test_poseidon_transcript_golden_cpp(H, golden):
    _setup_poseidon(H)
    for case in golden["poseidon_cases"]:
        rc, out = run_cpp_kind(H, 1, bytes.fromhex(case["script"]), bytes.fromhex(case["proof"]))
        assert rc == case["rc"], case["name"]
        assert out.hex() == case["out"], case["name"]


def test_poseidon_transcript_random_scripts_cpp_vs_oracle(H):
    import importlib.util

    _setup_poseidon(H)
    spec = importlib.util.spec_from_file_location("gen_t", os.path.join(ROOT, "tests", "golden", "gen_golden_transcript.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rng = random.Random(78)
    pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(5)] + [None]
    for _ in range(40):
        ops = []
        proof = b""
        for _ in range(rng.randrange(1, 14)):
            op = rng.choice([1, 2, 3, 4, 5, 6, 7, 8])
            if op in (2, 6):
                ops.append((op, rng.choice([0, 1, O.R - 1, rng.randrange(O.R)])))
            elif op in (3, 7):
                ops.append((op, rng.choice(pts)))
            else:
                ops.append((op, None))
            if op == 4:
                v = rng.randrange(O.R) if rng.random() < 0.9 else O.R + 5
                proof += v.to_bytes(32, "little")
            elif op == 5:
                proof += T.g1_compress(rng.choice(pts)) if rng.random() < 0.9 else (7).to_bytes(32, "little")
        exp = gen.run_script(ops, proof, kind=1)
        assert run_cpp_kind(H, 1, gen.pack_script(ops), proof) == exp
        assert run_cpp_kind(H, 2, gen.pack_script(ops), proof) == exp  # the eager sponge: same challenges, same stream


def test_poseidon_eager_sponge_equals_the_buffered_one_on_long_absorbs(H):
    """transcript.hpp `Poseidon::set_eager`: complete chunks of RATE elements are permuted as they arrive instead of inside
    `squeeze` (what lets the accumulation transcript of a pipelined aggregation job run under the work that feeds it,
    aggregation.hpp).  Every absorb count mod RATE before a squeeze, squeezes back to back, absorbs after a squeeze:
    the same challenges as the buffered sponge (the reference's shape, poseidon.rs:145-164) and as the oracle."""
    import importlib.util

    _setup_poseidon(H)
    spec = importlib.util.spec_from_file_location("gen_t", os.path.join(ROOT, "tests", "golden", "gen_golden_transcript.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rng = random.Random(2026)
    pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(4)]
    for case in range(12):
        ops = []
        for seg in range(rng.randrange(1, 5)):
            for _ in range(rng.choice([0, 1, 3, 4, 5, 8, 9, 17, 40])):
                if rng.random() < 0.5:
                    ops.append((2, rng.randrange(O.R)))
                else:
                    ops.append((3, rng.choice(pts)))
            ops.append((1, None))
            if rng.random() < 0.3:
                ops.append((1, None))
        script = gen.pack_script(ops)
        buffered, eager = run_cpp_kind(H, 1, script, b""), run_cpp_kind(H, 2, script, b"")
        assert buffered == eager and buffered[0] == 0 and len(buffered[1]) >= 32
        if case < 3:
            assert buffered == gen.run_script(ops, b"", kind=1)


def test_compressed_point_roundtrip():
    rng = random.Random(5)
    for _ in range(20):
        p = O.g1_mul(O.G1_GEN, rng.randrange(1, O.R))
        assert T.g1_decompress(T.g1_compress(p)) == p
        assert T.g1_decompress(T.g1_compress(O.g1_neg(p))) == O.g1_neg(p)
    assert T.g1_decompress(T.g1_compress(None)) is None


@pytest.mark.parametrize("params", [(3, 8, 57), (5, 8, 60), (2, 8, 56), (4, 8, 56), (6, 8, 57), (5, 4, 3)])
def test_poseidon_optimised_schedule_equals_plain_permutation(H, params):
    """transcript.hpp runs the optimised schedule (constants behind the S-boxes, sparse
    MDS in the partial rounds -- the shape of the reference's `permutation`, poseidon.rs:166-201);
    it must give the same state as the textbook rounds AND the oracle for any input."""
    _setup_poseidon(H)
    H.hd_poseidon_permute2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    t, rf, rp = params
    rng = random.Random(sum(params))
    for st in ([0] * t, [O.R - 1] * t, [rng.randrange(O.R) for _ in range(t)], [rng.randrange(O.R) for _ in range(t)]):
        outs = []
        for plain in (1, 0):
            buf = ctypes.create_string_buffer(b"".join(O.fe_to_bytes(x) for x in st), 32 * t)
            assert H.hd_poseidon_permute2(t, rf, rp, plain, buf) == 0
            outs.append([int.from_bytes(buf.raw[32 * i:32 * i + 32], "little") for i in range(t)])
        assert outs[0] == outs[1] == T.poseidon_permute(list(st), rf, rp)


def test_cost_estimation_of_the_standard_plonk_shape(H):
    """`CostEstimation` (cost.rs, verifier/plonk.rs:149-188, gwc19.rs:168-175, bdfg21.rs:379-385) on
    the StandardPlonk-shaped protocol of tests/plonk_synth.py: 6 witnesses + 3 quotient chunks,
    19 evaluations, rotations {0, 1, -1, last} -> 4 GWC19 opening commitments, 2 for BDFG21; 2 pairings."""
    import sys as _sys

    _sys.path.insert(0, os.path.join(ROOT, "tests"))
    import plonk_synth as S

    H.hd_plonk_estimate_cost.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    pr, _ = S.standard_plonk_protocol(random.Random(1), num_instance=(2,), accumulator_rows=None)
    pb = S.pack_protocol(pr)
    out = (ctypes.c_uint64 * 5)()
    assert H.hd_plonk_estimate_cost(0, pb, len(pb), out) == 0
    assert list(out) == [2, 9 + 4, 19, 8 + 9 + 1 + 4, 2]
    assert H.hd_plonk_estimate_cost(1, pb, len(pb), out) == 0
    assert list(out) == [2, 9 + 2, 19, 8 + 9 + 1 + 2, 2]


def test_poseidon_transcript_layout_record_rebuilds_every_absorbed_element(H):
    """transcript.hpp `record_layout`: the provenance a batch of proofs is described with to `snarkv_poseidon_read_batch`
    (lead value / scalar at a byte of the proof / coordinate of point q).  Random scripts: the C++ hook rebuilds every
    absorbed element from its code and compares with what the sponge got; here the codes, point offsets and segment
    lengths are checked against the script itself."""
    import importlib.util
    import struct

    spec = importlib.util.spec_from_file_location("gen_t", os.path.join(ROOT, "tests", "golden", "gen_golden_transcript.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    H.hd_poseidon_layout_script.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                            ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    rng = random.Random(4242)
    pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(6)]
    for _ in range(60):
        ops, proof = [], b""
        want_codes, want_offs, want_seg, cur, lead, q = [], [], [], 0, 0, 0
        for _ in range(rng.randrange(1, 30)):
            op = rng.choice([1, 2, 3, 4, 4, 5, 5])
            if op == 1:
                ops.append((1, None)); want_seg.append(cur); cur = 0
            elif op == 2:
                ops.append((2, rng.choice([0, 1, O.R - 1, rng.randrange(O.R)]))); want_codes.append((0 << 28) | lead); lead += 1; cur += 1
            elif op == 3:
                ops.append((3, rng.choice(pts))); want_codes += [(0 << 28) | lead, (0 << 28) | (lead + 1)]; lead += 2; cur += 2
            elif op == 4:
                ops.append((4, None)); want_codes.append((1 << 28) | len(proof)); cur += 1
                proof += rng.choice([0, O.R - 1, rng.randrange(O.R)]).to_bytes(32, "little")
            else:
                ops.append((5, None)); want_codes += [(2 << 28) | q, (3 << 28) | q]; q += 1; cur += 2
                want_offs.append(len(proof))
                proof += T.g1_compress(rng.choice(pts))
        out = ctypes.create_string_buffer(1 << 14)
        n = ctypes.c_size_t(0)
        script = gen.pack_script(ops)
        assert H.hd_poseidon_layout_script(script, len(script), proof, len(proof), out, len(out), ctypes.byref(n)) == 0
        w = struct.unpack("<%dI" % (n.value // 4), out.raw[:n.value])
        L = w[0]
        codes = list(w[1:1 + L])
        P = w[1 + L]
        offs = list(w[2 + L:2 + L + P])
        S = w[2 + L + P]
        seg = list(w[3 + L + P:3 + L + P + S])
        assert (codes, offs, seg) == (want_codes, want_offs, want_seg)
    # a scalar out of range / an invalid point stop the recording pass like any transcript (the fused route then never starts)
    bad = gen.pack_script([(4, None)])
    out = ctypes.create_string_buffer(64)
    n = ctypes.c_size_t(0)
    assert H.hd_poseidon_layout_script(bad, len(bad), (O.R + 1).to_bytes(32, "little"), 32, out, len(out), ctypes.byref(n)) == 1000


def test_ifma_permutation_equals_the_scalar_schedule_and_the_oracle(H):
    """host/poseidon_ifma.hpp (the state across AVX-512 lanes, 52-bit limbs, R = 2^260, no conditional subtraction on the
    way) against the scalar schedule and the Python oracle: random and extreme states, both parameter sets.  Skips where
    the CPU has no AVX-512 IFMA (the sponge then runs the scalar schedule, which the tests above pin)."""
    _setup_poseidon(H)
    H.hd_poseidon_permute_ifma.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    probe = ctypes.create_string_buffer(bytes(96), 96)
    if H.hd_poseidon_permute_ifma(3, 8, 57, probe) == 1:
        pytest.skip("no AVX-512 IFMA on this CPU")
    rng = random.Random(52)
    for (t, rf, rp) in ((3, 8, 57), (5, 8, 60)):
        states = [[0] * t, [O.R - 1] * t, [1] + [0] * (t - 1), [O.R - 1 - i for i in range(t)]]
        states += [[rng.randrange(O.R) for _ in range(t)] for _ in range(200)]
        for k, st in enumerate(states):
            a = ctypes.create_string_buffer(b"".join(O.fe_to_bytes(x) for x in st), 32 * t)
            assert H.hd_poseidon_permute(t, rf, rp, a) == 0
            # both forms of the partial round: the three-product form (default; words 1 .. t-1 renormalised every fourth
            # round) and the four-product form it replaced (still the route of a state of eight words)
            for form in (3, 4):
                assert H.hd_poseidon_ifma_form(form) in (3, 4)
                b = ctypes.create_string_buffer(b"".join(O.fe_to_bytes(x) for x in st), 32 * t)
                assert H.hd_poseidon_permute_ifma(t, rf, rp, b) == 0
                assert a.raw == b.raw, (t, k, form)
            H.hd_poseidon_ifma_form(3)
            if k < 8:
                got = [int.from_bytes(b.raw[32 * i:32 * i + 32], "little") for i in range(t)]
                assert got == T.poseidon_permute(st, rf, rp)


def test_poseidon_transcripts_on_the_scalar_schedule_too():
    """the sponge picks the IFMA path at run time; the same transcript tests with SNARKV_HOST_NO_IFMA=1 keep the scalar
    schedule covered on a CPU that has IFMA (a child interpreter: the choice is made once per process)"""
    import subprocess
    import sys

    if os.environ.get("SNARKV_HOST_NO_IFMA"):
        pytest.skip("already the scalar run")
    env = dict(os.environ, SNARKV_HOST_NO_IFMA="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-k", "poseidon and not scalar_schedule", "-p",
                        "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_device_point_hints_are_checked_and_strict_mode_refuses_a_disagreement(H):
    """`PoseidonTranscript::read_ec_point` with a device-decoded hint beside the bytes (host/transcript.hpp): a hint is used
    only if it IS the decoding of the 32 bytes; an unusable hint falls back to the host's own decoding -- EXCEPT in strict
    mode (the fused device route, whose challenges were hashed over the device's decodings): there a finite point the host
    decodes although the device's answer was unusable is a disagreement and must be refused."""
    H.hd_poseidon_hint_policy.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    rng = random.Random(91)
    p = O.g1_mul(O.G1_GEN, rng.randrange(1, O.R))
    q = O.g1_mul(O.G1_GEN, rng.randrange(1, O.R))
    enc, pb, qb = T.g1_compress(p), O.g1_to_bytes(p), O.g1_to_bytes(q)
    out = ctypes.create_string_buffer(64)
    for strict in (0, 1):
        assert H.hd_poseidon_hint_policy(enc, pb, 1, strict, out) == 1 and out.raw == pb  # the right hint: taken
    # a wrong point with ok = 1, or the right point flagged unusable: the host decodes itself ... unless strict
    for hint, ok in ((qb, 1), (pb, 0), (bytes(64), 1)):
        assert H.hd_poseidon_hint_policy(enc, hint, ok, 0, out) == 1 and out.raw == pb
        assert H.hd_poseidon_hint_policy(enc, hint, ok, 1, out) == 0
    # an invalid encoding is Error::Transcript either way (x = 7 is not on the curve: 343 + 3 is a non-residue? found by search)
    x = next(x for x in range(2, 50) if pow((x ** 3 + 3) % O.P, (O.P - 1) // 2, O.P) != 1)
    bad = x.to_bytes(32, "little")
    for strict in (0, 1):
        assert H.hd_poseidon_hint_policy(bad, bytes(64), 0, strict, out) == 0


def test_ifma_batched_point_decompression_is_a_checked_hint(H):
    """transcript.hpp `g1_decompress_x8`: up to eight compressed points decoded together on AVX-512 IFMA (one square-root
    chain for the group) -- what a Poseidon transcript does for the points of one `read_n_ec_points`.  Against the oracle's
    decompression: random points of both parities in every group size, and the encodings it must LEAVE to the scalar decoder
    (flag 0): the identity, x >= p, x with no square root on the curve, the identity flag over a non-zero x."""
    H.hd_g1_decompress_x8.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p]
    probe_out, probe_ok = ctypes.create_string_buffer(64), ctypes.create_string_buffer(1)
    if H.hd_g1_decompress_x8(T.g1_compress(O.G1_GEN), 1, probe_out, probe_ok) == 1:
        pytest.skip("no AVX-512 IFMA on this CPU")
    rng = random.Random(88)

    def non_residue_x():
        while True:
            x = rng.randrange(O.P)
            if pow((x * x * x + 3) % O.P, (O.P - 1) // 2, O.P) != 1:
                return x

    for n in list(range(1, 9)) * 6:
        encs, exp = [], []
        for _ in range(n):
            kind = rng.choice(["pt", "pt", "pt", "neg", "inf", "big", "nonres", "inf_garbage", "small"])
            if kind in ("pt", "neg", "small"):
                p = O.g1_mul(O.G1_GEN, rng.randrange(1, 50) if kind == "small" else rng.randrange(1, O.R))
                if kind == "neg":
                    p = O.g1_neg(p)
                encs.append(T.g1_compress(p))
                exp.append(p)
            elif kind == "inf":
                encs.append(T.g1_compress(None))
                exp.append(None)
            elif kind == "big":
                encs.append((O.P + rng.randrange(1 << 60)).to_bytes(32, "little"))
                exp.append(None)
            elif kind == "nonres":
                e = bytearray(non_residue_x().to_bytes(32, "little"))
                e[31] |= rng.randrange(2) << 6
                encs.append(bytes(e))
                exp.append(None)
            else:
                e = bytearray(rng.randrange(1, O.P).to_bytes(32, "little"))
                e[31] |= 0x80
                encs.append(bytes(e))
                exp.append(None)
        out, ok = ctypes.create_string_buffer(64 * n), ctypes.create_string_buffer(n)
        assert H.hd_g1_decompress_x8(b"".join(encs), n, out, ok) == 0
        for i in range(n):
            if exp[i] is None:
                assert ok.raw[i] == 0, (n, i)
            else:
                assert ok.raw[i] == 1, (n, i)
                assert out.raw[64 * i:64 * i + 64] == O.g1_to_bytes(exp[i]), (n, i)
                assert T.g1_decompress(encs[i]) == exp[i]
