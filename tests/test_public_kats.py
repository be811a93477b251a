"""Public known-answer vectors (tests/golden/public_kats.json: EIP-196 ecAdd / ecMul, EIP-197
pairing check, Keccak-256, circomlib Poseidon) through BOTH oracles (Python big-int, C 4x64
Montgomery), the C++ host mirror (Keccak, Poseidon) and -- under `-m gpu` -- the device path.

These are the only vectors here that were not produced by code in this repository: they pin the
oracle (and through it every parity test) to the published arithmetic of alt_bn128 / BN254, the
curve halo2curves' `bn256` implements (reference Cargo.toml:14)."""
import ctypes
import json
import os

import pytest

import bn254 as O
import coracle as C
import transcript as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "public_kats.json")) as f:
        return json.load(f)


def _pt(xy):
    x, y = int(xy[0], 16), int(xy[1], 16)
    return None if x == 0 and y == 0 else (x, y)  # the precompiles' (and the reference's) identity: (0, 0)


def _raw(xy):  # 64 bytes as the C ABI takes them, whether or not they are a valid point
    return bytes.fromhex(xy[0])[::-1] + bytes.fromhex(xy[1])[::-1]


def _pairs(words):
    v = [int(w, 16) for w in words]
    out = []
    for i in range(0, len(v), 6):  # EIP-197 encodes Fq2 as (imaginary, real)
        out.append(((v[i], v[i + 1]), (O.Fq2(v[i + 3], v[i + 2]), O.Fq2(v[i + 5], v[i + 4]))))
    return out


def test_eip196_python_and_c_oracles(kats):
    assert len(kats["eip196_ecadd"]) >= 8 and len(kats["eip196_ecmul"]) >= 10
    for c in kats["eip196_ecadd"]:
        p, q, s = _pt(c["p"]), _pt(c["q"]), _pt(c["sum"])
        assert O.g1_is_on_curve(p) and O.g1_is_on_curve(q) and O.g1_is_on_curve(s), c["name"]
        assert O.g1_add(p, q) == s, c["name"]
        assert C.g1_add(O.g1_to_bytes(p), O.g1_to_bytes(q)) == O.g1_to_bytes(s), c["name"]
        one = O.fe_to_bytes(1)
        assert C.msm_naive(one + one, O.g1_to_bytes(p) + O.g1_to_bytes(q)) == O.g1_to_bytes(s), c["name"]
        assert C.msm_pippenger(one + one, O.g1_to_bytes(p) + O.g1_to_bytes(q), 1) == O.g1_to_bytes(s), c["name"]
    for c in kats["eip196_ecmul"]:
        p, k, e = _pt(c["p"]), int(c["k"], 16), _pt(c["product"])
        assert O.g1_is_on_curve(p) and O.g1_is_on_curve(e), c["name"]
        assert O.g1_mul(p, k) == e, c["name"]  # the big-integer oracle takes the raw 256-bit scalar, as the precompile does
        kr = O.fe_to_bytes(k % O.R)            # `Fr` values are canonical: the C ABI's scalars are k mod r
        assert C.g1_mul(O.g1_to_bytes(p), kr) == O.g1_to_bytes(e), c["name"]
        assert C.msm_naive(kr, O.g1_to_bytes(p)) == O.g1_to_bytes(e), c["name"]
        assert C.msm_pippenger(kr, O.g1_to_bytes(p), 1) == O.g1_to_bytes(e), c["name"]
    for c in kats["eip196_invalid_points"]:
        x, y = int(c["xy"][0], 16), int(c["xy"][1], 16)
        assert not O.g1_is_on_curve((x, y)), c["name"]
        if x < O.P and y < O.P:
            assert not C.g1_is_on_curve(_raw(c["xy"])), c["name"]


def _two_pair_checks(c):
    """the case's pairs taken two at a time: [((P1, Q1), (P2, Q2), product_is_one)] by the big-integer oracle"""
    pairs = _pairs(c["words"])
    out = []
    for i in range(0, len(pairs) - 1, 2):
        two = pairs[i:i + 2]
        out.append((two[0], two[1], O.final_exponentiation(O.miller_loop(two)) == O.FQ12_ONE))
    return out


def test_eip197_pairing_check_python_oracle(kats):
    assert len(kats["eip197_pairing_check"]) >= 9
    for c in kats["eip197_pairing_check"]:
        pairs = _pairs(c["words"])
        for p, q in pairs:
            assert O.g1_is_on_curve(p) and O.g2_is_on_curve(q)
        assert (O.final_exponentiation(O.miller_loop(pairs)) == O.FQ12_ONE) == bool(c["result"]), c["name"]
        if c["name"] == "jeff1":
            assert pairs[1][1] == O.G2_GEN  # the EIP-197 generator the deciding keys of the tests are built on
        if len(pairs) == 2:
            # the decider form e(lhs, g2) e(rhs, -s_g2) = 1 (pcs/kzg/decider.rs:70-82) of the same check
            (p1, q1), (p2, q2) = pairs
            assert O.kzg_decide(p2, p1, q2, O.g2_neg(q1)) == bool(c["result"]), c["name"]


def test_eip197_pairing_check_c_oracle(kats):
    n = 0
    for c in kats["eip197_pairing_check"]:
        for (p1, q1), (p2, q2), ok in _two_pair_checks(c):
            acc = O.g1_to_bytes(p2) + O.g1_to_bytes(p1)
            assert C.kzg_decide(O.g2_to_bytes(q2), O.g2_to_bytes(O.g2_neg(q1)), acc) is ok, c["name"]
            n += 1
        if c["name"] == "jeff1":
            (p1, q1), (p2, q2) = _pairs(c["words"])
            bad = O.g1_to_bytes(O.g1_double(p2)) + O.g1_to_bytes(p1)
            assert C.kzg_decide(O.g2_to_bytes(q2), O.g2_to_bytes(O.g2_neg(q1)), bad) is False
    assert n >= 14


def test_keccak_and_poseidon_public_values(kats):
    H = ctypes.CDLL(os.path.join(ROOT, "snark-verifier_amd", "libsnarkv_hosttest.so"))
    H.hd_keccak256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    H.hd_poseidon_permute.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    for c in kats["keccak256"]:
        msg = bytes.fromhex(c["msg_hex"])
        assert T.keccak256(msg).hex() == c["digest"]
        out = ctypes.create_string_buffer(32)
        H.hd_keccak256(msg, len(msg), out)
        assert out.raw.hex() == c["digest"]
    for c in kats["poseidon_circomlib"]:
        st = [0] + c["inputs"]
        assert len(st) == c["t"]
        exp = int(c["hash_dec"])
        assert T.poseidon_permute(list(st), c["r_f"], c["r_p"])[0] == exp
        assert T.poseidon_permute_opt(list(st), c["t"], c["r_f"], c["r_p"])[0] == exp
        buf = ctypes.create_string_buffer(b"".join(O.fe_to_bytes(x) for x in st), 32 * c["t"])
        assert H.hd_poseidon_permute(c["t"], c["r_f"], c["r_p"], buf) == 0
        assert int.from_bytes(buf.raw[:32], "little") == exp
        # ... and the AVX-512 IFMA permutation the sponge runs where the CPU has it (rc 1: it has not)
        H.hd_poseidon_permute_ifma.argtypes = H.hd_poseidon_permute.argtypes
        buf = ctypes.create_string_buffer(b"".join(O.fe_to_bytes(x) for x in st), 32 * c["t"])
        rc = H.hd_poseidon_permute_ifma(c["t"], c["r_f"], c["r_p"], buf)
        assert rc in (0, 1) and (rc == 1 or int.from_bytes(buf.raw[:32], "little") == exp)


# ------------------------------------------------------------------ device
@pytest.mark.gpu
def test_eip196_on_device(gpu_ctx, kats):
    import snark_verifier_amd as sv

    one = O.fe_to_bytes(1)
    n = 0
    for c in kats["eip196_ecadd"]:
        p, q, s = (O.g1_to_bytes(_pt(c[k])) for k in ("p", "q", "sum"))
        assert gpu_ctx.msm_naive(one + one, p + q) == s, c["name"]
        assert gpu_ctx.msm_pippenger(one + one, p + q) == s, c["name"]
        assert gpu_ctx.msm_batched(one + one, p + q, [0, 1, 2]) == p + q, c["name"]
        assert gpu_ctx.msm_naive(one + one, p + q, sv.SNARKV_FLAG_VALIDATE) == s, c["name"]  # valid encodings, (0, 0) included
        n += 1
    for c in kats["eip196_ecmul"]:
        p, e = O.g1_to_bytes(_pt(c["p"])), O.g1_to_bytes(_pt(c["product"]))
        k = O.fe_to_bytes(int(c["k"], 16) % O.R)  # canonical `Fr`, as the reference's types guarantee at this boundary
        assert gpu_ctx.msm_naive(k, p) == e, c["name"]
        assert gpu_ctx.msm_pippenger(k, p) == e, c["name"]
        n += 1
    assert n >= 18
    # the precompiles' failure cases: off-curve / non-canonical encodings are rejected under the validate flag
    for c in kats["eip196_invalid_points"]:
        for fn in (gpu_ctx.msm_naive, gpu_ctx.msm_pippenger):
            with pytest.raises(sv.SnarkvError) as e:
                fn(one, _raw(c["xy"]), sv.SNARKV_FLAG_VALIDATE)
            assert e.value.code == sv.SNARKV_ERR_ENCODING, c["name"]
    with pytest.raises(sv.SnarkvError) as e:  # scalar = r: not a canonical Fr
        gpu_ctx.msm_naive(O.fe_to_bytes(O.R), O.g1_to_bytes(O.G1_GEN), sv.SNARKV_FLAG_VALIDATE)
    assert e.value.code == sv.SNARKV_ERR_ENCODING


@pytest.mark.gpu
@pytest.mark.parametrize("teams", ["1", "3"])
def test_eip197_pairing_check_on_device(gpu_ctx, kats, teams, monkeypatch):
    """e(P1,Q1) e(P2,Q2) = 1 fed to the HIP decider as e(lhs, g2) e(rhs, -s_g2) with g2 := Q2,
    s_g2 := -Q1, lhs := P2, rhs := P1, for both kernel forms; cases of more than two pairs two pairs at a time
    (every sub-product against the big-integer oracle)."""
    import snark_verifier_amd as sv

    monkeypatch.setenv("SNARKV_DECIDE_FORM", teams)
    n = 0
    for c in kats["eip197_pairing_check"]:
        for (p1, q1), (p2, q2), ok in _two_pair_checks(c):
            dk = sv.DecidingKey(gpu_ctx, O.g1_to_bytes(O.G1_GEN), O.g2_to_bytes(q2), O.g2_to_bytes(O.g2_neg(q1)))
            acc = O.g1_to_bytes(p2) + O.g1_to_bytes(p1)
            assert gpu_ctx.decide(dk, acc) is ok, c["name"]
            n += 1
            if c["name"] == "jeff1":
                bad = O.g1_to_bytes(O.g1_double(p2)) + O.g1_to_bytes(p1)
                assert gpu_ctx.decide(dk, bad) is False
                allok, oks = gpu_ctx.decide_batch(dk, acc + bad + acc)
                assert not allok and oks == [True, False, True]
            dk.close()
    assert n >= 14
