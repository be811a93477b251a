"""GPU: the C++ mirror of the reference's IPA layer (snark-verifier_amd/host/ipa.hpp) against the
oracle: `Ipa::succinct_verify` (two MSMs, one segmented launch), `IpaAs::verify`, and
`IpaAs::decide` -- ONE device Pippenger of 2^k terms, the reference's second consumer of
`util::msm::multi_scalar_multiplication` (pcs/ipa/decider.rs:51-52)."""
import ctypes
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bn254 as O  # noqa: E402
import coracle as C  # noqa: E402
import ipa as I  # noqa: E402
import transcript as T  # noqa: E402
from ipa_util import acc_from_json, bgh19_case, case_key, load_cases, pack_acc, pack_svk  # noqa: E402

pytestmark = pytest.mark.gpu
TR = {"evm": (0, T.EvmTranscript), "poseidon": (1, T.PoseidonTranscript)}


@pytest.fixture(scope="module")
def H():
    from hostfmt import load_host_lib

    h = load_host_lib()
    cp, u32, sz = ctypes.c_char_p, ctypes.c_uint32, ctypes.c_size_t
    h.hd_ipa_succinct_verify.argtypes = [ctypes.c_int, cp, cp, cp, cp, cp, sz, cp]
    h.hd_ipa_as_verify.argtypes = [ctypes.c_int, cp, cp, u32, cp, sz, cp]
    h.hd_ipa_decide_all.argtypes = [u32, cp, sz, cp, u32]
    h.hd_ipa_h.argtypes = [u32, cp, cp, cp]
    h.hd_ipa_bgh19_verify.argtypes = [ctypes.c_int, cp, cp, cp, cp, cp, sz, cp]
    return h


def _buf(n):
    return ctypes.create_string_buffer(n)


def test_h_eval_and_h_coeffs(H):
    rnd = random.Random(3)
    for k in (1, 2, 6, 13):  # 13: the threaded path of h_coeffs
        xi = [rnd.randrange(O.R) for _ in range(k)]
        z = rnd.randrange(O.R)
        out = _buf(32 + 32 * (1 << k))
        assert H.hd_ipa_h(k, b"".join(O.fe_to_bytes(x) for x in xi), O.fe_to_bytes(z), out) == 0
        assert out.raw[:32] == O.fe_to_bytes(I.h_eval(xi, z))
        assert out.raw[32:] == b"".join(O.fe_to_bytes(c) for c in I.h_coeffs(xi, 1))
    assert H.hd_ipa_h(0, b"", O.fe_to_bytes(1), _buf(64)) == -100  # ipa.rs:406 assert -> panic


@pytest.mark.parametrize("idx", range(4))
def test_golden_succinct_verify_accumulate_decide(H, idx):
    c = load_cases()[idx]
    g, h, s = case_key(c)
    tk, _ = TR[c["transcript"]]
    k = c["k"]
    svk = pack_svk(k, g[0], h, s)
    gb = b"".join(O.g1_to_bytes(p) for p in g)
    stride = 32 * k + 64
    accs = []
    for o in c["openings"]:
        proof = bytes.fromhex(o["proof"])
        com, z, ev = (bytes.fromhex(o[n]) for n in ("commitment", "z", "eval"))
        out = _buf(stride)
        assert H.hd_ipa_succinct_verify(tk, svk, com, z, ev, proof, len(proof), out) == 1
        assert out.raw == pack_acc(acc_from_json(o["accumulator"]))
        accs.append(out.raw)
        bad_ev = O.fe_to_bytes((O.fe_from_bytes(ev) + 1) % O.R)
        assert H.hd_ipa_succinct_verify(tk, svk, com, z, bad_ev, proof, len(proof), out) == 0   # AssertionFailure
        assert H.hd_ipa_succinct_verify(tk, svk, com, z, ev, proof[:-1], len(proof) - 1, out) == -10  # Transcript
        flipped = bytearray(proof)
        flipped[len(proof) // 2] ^= 1
        assert H.hd_ipa_succinct_verify(tk, svk, com, z, ev, bytes(flipped), len(proof), out) in (0, -10)
    assert H.hd_ipa_decide_all(k, gb, len(g), b"".join(accs), len(accs)) == 1
    out = _buf(stride)
    as_proof = bytes.fromhex(c["as_proof"])
    assert H.hd_ipa_as_verify(tk, svk, b"".join(accs), len(accs), as_proof, len(as_proof), out) == 1
    assert out.raw == pack_acc(acc_from_json(c["as_accumulator"]))
    assert H.hd_ipa_decide_all(k, gb, len(g), out.raw, 1) == 1
    # decide rejects a perturbed U / xi, wherever it sits in the list
    xi, u = acc_from_json(c["as_accumulator"])
    bad_u = pack_acc((xi, O.g1_add(u, h)))
    bad_xi = pack_acc(([(xi[0] + 1) % O.R] + xi[1:], u))
    assert H.hd_ipa_decide_all(k, gb, len(g), bad_u, 1) == 0
    assert H.hd_ipa_decide_all(k, gb, len(g), bad_xi, 1) == 0
    assert H.hd_ipa_decide_all(k, gb, len(g), b"".join(accs) + bad_u, len(accs) + 1) == 0
    # one old accumulator replaced: the honest IpaAs proof no longer verifies
    assert H.hd_ipa_as_verify(tk, svk, accs[0] + bad_u[:stride] + accs[2], 3, as_proof, len(as_proof), out) in (0, -10)
    # a single instance: the reference asserts (accumulation.rs:107)
    assert H.hd_ipa_as_verify(tk, svk, accs[0], 1, as_proof, len(as_proof), out) == -100
    # committing key of the wrong length: `assert_eq!(scalars.len(), bases.len())` (msm.rs:309)
    assert H.hd_ipa_decide_all(k, gb[:-64], len(g) - 1, accs[0], 1) == -100


@pytest.mark.parametrize("k,zk", [(10, False), (10, True)])
def test_reference_test_sizes(H, k, zk):
    """`test_ipa` (pcs/ipa.rs:434-466: k = 10, zk in {false, true}) and `test_ipa_as`
    (accumulation.rs:240-290) with a seeded RNG: oracle prover -> C++ verifier -> device decide."""
    rnd = random.Random("ref-%d-%d" % (k, zk))
    rng = lambda: rnd.randrange(O.R)  # noqa: E731
    n = 1 << k
    raw = C.sample_points(1000 + k + zk, n + 2)
    pts = [O.g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range(n + 2)]
    pk = I.IpaProvingKey(k, pts[:n], pts[n], pts[n + 1] if zk else None)
    svk, gb, stride = pack_svk(k, pk.g[0], pk.h, pk.s), raw[:64 * n], 32 * k + 64
    accs, accs_py = [], []
    for _ in range(3):
        p = [rng() for _ in range(n)]
        omega, z = (rng() if zk else None), rng()
        com = pk.commit(p, omega)
        t = T.EvmTranscript()
        acc = I.ipa_create_proof(pk, p, z, omega, t, rng)
        proof = t.finalize()
        out = _buf(stride)
        assert H.hd_ipa_succinct_verify(0, svk, O.g1_to_bytes(com), O.fe_to_bytes(z), O.fe_to_bytes(I.poly_eval(p, z)),
                                        proof, len(proof), out) == 1
        assert out.raw == pack_acc(acc)
        accs.append(out.raw)
        accs_py.append(acc)
    assert H.hd_ipa_decide_all(k, gb, n, b"".join(accs), 3) == 1
    t = T.EvmTranscript()
    acc = I.ipa_as_create_proof(pk, accs_py, t, rng)
    as_proof = t.finalize()
    out = _buf(stride)
    assert H.hd_ipa_as_verify(0, svk, b"".join(accs), 3, as_proof, len(as_proof), out) == 1
    assert out.raw == pack_acc(acc)
    assert H.hd_ipa_decide_all(k, gb, n, out.raw, 1) == 1
    bad = bytearray(out.raw)
    bad[5] ^= 4
    assert H.hd_ipa_decide_all(k, gb, n, bytes(bad), 1) == 0


def test_decide_is_one_large_msm_2p16(H):
    """decide at k = 16: 65 536-term device Pippenger against the C oracle's MSM of the same h_coeffs."""
    k, n = 16, 1 << 16
    rnd = random.Random(9)
    xi = [rnd.randrange(O.R) for _ in range(k)]
    gb = C.sample_points(4242, n)
    hb = b"".join(O.fe_to_bytes(c) for c in I.h_coeffs(xi, 1))
    u = C.msm_pippenger(hb, gb, 8)
    acc = b"".join(O.fe_to_bytes(x) for x in xi) + u
    assert H.hd_ipa_decide_all(k, gb, n, acc, 1) == 1
    other = C.g1_add(u, gb[:64])
    assert H.hd_ipa_decide_all(k, gb, n, acc[:32 * k] + other, 1) == 0


@pytest.mark.parametrize("idx", range(2))
def test_golden_bgh19_multiopen(H, idx):
    """`IpaAs<Bgh19>` read_proof + verify (multiopen/bgh19.rs:26-153) in the C++ mirror: same
    accumulator bytes as the oracle, decide accepts; wrong evaluations / commitments reject."""
    import kzg as K
    from hostfmt import pack_commitments, pack_queries

    c = load_cases("bgh19")[idx]
    g, h, s, coms, x, queries, proof, exp = bgh19_case(c)
    tk, k = TR[c["transcript"]][0], c["k"]
    svk = pack_svk(k, g[0], h, s)
    cm = pack_commitments([K.Msm.base(p) for p in coms])
    out = _buf(32 * k + 64)
    assert H.hd_ipa_bgh19_verify(tk, svk, cm, O.fe_to_bytes(x), pack_queries(queries), proof, len(proof), out) == 1
    assert out.raw == pack_acc(exp)
    gb = b"".join(O.g1_to_bytes(p) for p in g)
    assert H.hd_ipa_decide_all(k, gb, len(g), out.raw, 1) == 1
    for i in (0, 5, 8):
        bad = list(queries)
        bad[i] = (bad[i][0], bad[i][1], (bad[i][2] + 1) % O.R)
        assert H.hd_ipa_bgh19_verify(tk, svk, cm, O.fe_to_bytes(x), pack_queries(bad), proof, len(proof), out) == 0
    cm2 = pack_commitments([K.Msm.base(p) for p in [coms[1], coms[0]] + coms[2:]])
    assert H.hd_ipa_bgh19_verify(tk, svk, cm2, O.fe_to_bytes(x), pack_queries(queries), proof, len(proof), out) == 0
    assert H.hd_ipa_bgh19_verify(tk, svk, cm, O.fe_to_bytes(x), pack_queries(queries), proof[:-7], len(proof) - 7, out) == -10
    # commitments given as linear combinations (what the PLONK verifier hands over): 2*C0 - C0 = C0
    lin = [K.Msm.base(coms[0]) * 2 - K.Msm.base(coms[0])] + [K.Msm.base(p) for p in coms[1:]]
    assert H.hd_ipa_bgh19_verify(tk, svk, pack_commitments(lin), O.fe_to_bytes(x), pack_queries(queries), proof,
                                 len(proof), out) == 1
    assert out.raw == pack_acc(exp)


# ---- PLONK over IPA: `PlonkVerifier<IpaAs<Bgh19>>` (the reference's system/halo2/test/ipa/native.rs) ----
def _plonk_ipa_setup(seed, k, **kw):
    import plonk as P
    import plonk_synth as S

    rng = random.Random(seed)
    pr, dl = S.standard_plonk_protocol(rng, k=k, **kw)
    inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
    kd = {"g": [rng.randrange(1, O.R) for _ in range(1 << k)], "h": rng.randrange(1, O.R), "s": rng.randrange(1, O.R)}
    gb = b"".join(C.g1_mul(O.g1_to_bytes(O.G1_GEN), O.fe_to_bytes(c)) for c in kd["g"])
    g = [O.g1_from_bytes(gb[64 * i:64 * i + 64]) for i in range(1 << k)]
    h, s = O.g1_mul(O.G1_GEN, kd["h"]), O.g1_mul(O.G1_GEN, kd["s"])
    return P, S, rng, pr, dl, inst, kd, g, gb, h, s


def _plonk_ipa_run(H, tk, S, pr, inst, proof, svk, gb, k, decide=1):
    H.hd_plonk_ipa_verify.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                      ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                                      ctypes.c_char_p, ctypes.c_int]
    pb, ib = S.pack_protocol(pr), S.pack_instances(inst)
    out = _buf(32 * k + 64)
    rc = H.hd_plonk_ipa_verify(tk, pb, len(pb), ib, len(ib), proof, len(proof), svk, gb, len(gb) // 64, out, decide)
    return rc, out.raw


@pytest.mark.parametrize("kind", ["evm", "poseidon"])
@pytest.mark.parametrize("lin", [None, "WithoutConstant", "MinusVanishingTimesQuotient"])
def test_plonk_over_ipa_forged_proofs(H, kind, lin):
    """Proofs forged under a committing key with known discrete logs (oracle/plonk.py
    `forge_proof_ipa`): the C++ `PlonkSuccinctVerifier<Bgh19>` returns the oracle's accumulator bytes,
    `PlonkVerifier<Bgh19>::verify` accepts (decide = one device MSM), any change rejects."""
    k = 5
    P, S, rng, pr, dl, inst, kd, g, gb, h, s = _plonk_ipa_setup("pipa-%s-%s" % (kind, lin), k, linearization=lin,
                                                                num_instance=(2, 3))
    tk, Tr = TR[kind]
    proof = P.forge_proof_ipa(pr, inst, kd, Tr, rng, dl)
    exp = P.succinct_verify_ipa(g[0], h, s, pr, inst, P.plonk_proof_read(pr, inst, Tr(proof), "bgh19"))
    assert I.ipa_decide(g, exp[0])
    svk = pack_svk(k, g[0], h, s)
    rc, acc = _plonk_ipa_run(H, tk, S, pr, inst, proof, svk, gb, k)
    assert rc == 1 and acc == pack_acc(exp[0])
    for pos in (0, len(proof) // 2, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert _plonk_ipa_run(H, tk, S, pr, inst, bytes(bad), svk, gb, k)[0] in (0, -10)
    inst2 = [list(x) for x in inst]
    inst2[1][2] = (inst2[1][2] + 1) % O.R
    assert _plonk_ipa_run(H, tk, S, pr, inst2, proof, svk, gb, k)[0] == 0
    assert _plonk_ipa_run(H, tk, S, pr, [inst[0]], proof, svk, gb, k)[0] == -11  # InvalidInstances
    # a committing key that differs in ONE point: the succinct check still passes (it never sees G_i, i > 0),
    # `decide` is what catches it
    gb2 = gb[:64 * 7] + gb[64 * 8:64 * 9] + gb[64 * 8:]
    assert _plonk_ipa_run(H, tk, S, pr, inst, proof, svk, gb2, k, decide=0)[0] == 1
    assert _plonk_ipa_run(H, tk, S, pr, inst, proof, svk, gb2, k, decide=1)[0] == 0


def test_plonk_over_ipa_rejects_accumulator_indices(H):
    """IPA has no `AccumulatorEncoding`: the reference's default (`PhantomData`, pcs.rs:173-184) is
    `unimplemented!()`, so a protocol that declares old accumulators panics."""
    k = 4
    P, S, rng, pr, dl, inst, kd, g, gb, h, s = _plonk_ipa_setup(99, k, num_instance=(17,),
                                                                accumulator_rows=[list(range(16))])
    pr2 = dict(pr, accumulator_indices=[])
    proof = P.forge_proof_ipa(pr2, inst, kd, T.EvmTranscript, rng, dl)
    svk = pack_svk(k, g[0], h, s)
    assert _plonk_ipa_run(H, 0, S, pr2, inst, proof, svk, gb, k)[0] == 1
    assert _plonk_ipa_run(H, 0, S, pr, inst, proof, svk, gb, k)[0] == -100


# ---- the C-ABI decider itself (include/snarkv_amd.h `snarkv_ipa_*`) -----------------------------
def test_c_abi_ipa_decide_batch_against_the_oracle(gpu_ctx):
    import snark_verifier_amd as sv

    rnd = random.Random(21)
    for k in (1, 2, 7, 12):
        n = 1 << k
        gb = C.sample_points(900 + k, n)
        dk = sv.IpaDecidingKey(gpu_ctx, gb)
        assert dk.k == k
        xis, us, want = [], [], []
        for a in range(5):
            xi = [rnd.randrange(O.R) for _ in range(k)]
            if a == 3:
                xi[0] = 0  # a zero challenge: half of h_coeffs vanish (zero scalars in the MSM)
            hb = b"".join(O.fe_to_bytes(c) for c in I.h_coeffs(xi, 1))
            u = C.msm_pippenger(hb, gb, 4)
            good = a != 1
            if not good:
                u = C.g1_add(u, gb[:64])
            xis.append(b"".join(O.fe_to_bytes(x) for x in xi))
            us.append(u)
            want.append(good)
        assert gpu_ctx.ipa_decide_batch(dk, b"".join(xis), b"".join(us)) == want
        dk.close()


def test_c_abi_ipa_error_codes(gpu_ctx):
    import snark_verifier_amd as sv

    gb = C.sample_points(5, 8)
    with pytest.raises(sv.SnarkvError):
        sv.IpaDecidingKey(gpu_ctx, gb[:64 * 6])  # not a power of two
    with pytest.raises(sv.SnarkvError):
        sv.IpaDecidingKey(gpu_ctx, b"")          # empty
    dk = sv.IpaDecidingKey(gpu_ctx, gb)
    with pytest.raises(sv.SnarkvError):
        gpu_ctx.ipa_decide_batch(dk, b"", b"")   # m = 0


def test_c_abi_ipa_decide_2p20_resident_key(gpu_ctx):
    """k = 20: the committing key stays on the device, a decide uploads 20 scalars.  Checked through
    the homomorphism instead of a CPU MSM of 2^20 terms: with G' = (G, G) (the key repeated),
    commit_{G'}(h(xi_1..xi_k)) = (1 + xi_1) * commit_G(h(xi_2..xi_k))."""
    import snark_verifier_amd as sv

    k = 19
    n = 1 << k
    gb = C.sample_points(31337, n)
    rnd = random.Random(4)
    xi = [rnd.randrange(O.R) for _ in range(k)]
    hb = b"".join(O.fe_to_bytes(c) for c in I.h_coeffs(xi, 1))
    out = gpu_ctx.msm_pippenger(hb, gb)
    x0 = rnd.randrange(O.R)
    u2 = C.g1_mul(out, O.fe_to_bytes((1 + x0) % O.R))
    dk2 = sv.IpaDecidingKey(gpu_ctx, gb + gb)
    assert dk2.k == k + 1
    xi2 = b"".join(O.fe_to_bytes(x) for x in [x0] + xi)
    assert gpu_ctx.ipa_decide_batch(dk2, xi2 + xi2, u2 + out) == [True, False]


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_protocol_shapes_over_ipa(H, seed):
    """Fuzzing `PlonkVerifier<IpaAs<Bgh19>>` in the C++ mirror against the oracle: random polynomial
    counts, phases, rotations (so random Bgh19 rotation sets), expression trees, quotient chunking,
    linearization modes, both transcripts; k = 3..7."""
    import plonk as P
    import plonk_synth as S

    rng = random.Random(9000 + seed)
    lin = rng.choice([None, None, "WithoutConstant", "MinusVanishingTimesQuotient"])
    kind = rng.choice(["evm", "poseidon"])
    pr, dl = S.random_protocol(rng, lin)
    k = pr["domain"].k
    inst = [[rng.randrange(O.R) for _ in range(n)] for n in pr["num_instance"]]
    kd = {"g": [rng.randrange(1, O.R) for _ in range(1 << k)], "h": rng.randrange(1, O.R), "s": rng.randrange(1, O.R)}
    gb = b"".join(C.g1_mul(O.g1_to_bytes(O.G1_GEN), O.fe_to_bytes(c)) for c in kd["g"])
    g = [O.g1_from_bytes(gb[64 * i:64 * i + 64]) for i in range(1 << k)]
    h, s = O.g1_mul(O.G1_GEN, kd["h"]), O.g1_mul(O.G1_GEN, kd["s"])
    tk, Tr = TR[kind]
    proof = P.forge_proof_ipa(pr, inst, kd, Tr, rng, dl)
    exp = P.succinct_verify_ipa(g[0], h, s, pr, inst, P.plonk_proof_read(pr, inst, Tr(proof), "bgh19"))
    rc, acc = _plonk_ipa_run(H, tk, S, pr, inst, proof, pack_svk(k, g[0], h, s), gb, k)
    assert rc == 1, (seed, lin, kind, k)
    assert acc == pack_acc(exp[0])
    bad = bytearray(proof)
    bad[rng.randrange(len(proof))] ^= 1 << rng.randrange(8)
    assert _plonk_ipa_run(H, tk, S, pr, inst, bytes(bad), pack_svk(k, g[0], h, s), gb, k)[0] in (0, -10)


@pytest.mark.parametrize("seed", list(range(6)))
def test_random_bgh19_query_patterns_with_the_polynomial_prover(H, seed):
    """Honest Bgh19 openings (oracle prover over real polynomials) of random query patterns: random
    numbers of polynomials, rotations drawn from {0, 1, -1, 2, -3}, duplicated queries."""
    import kzg as K
    from hostfmt import pack_commitments, pack_queries

    rng = random.Random(7000 + seed)
    rnd = lambda: rng.randrange(O.R)  # noqa: E731
    k = rng.randrange(2, 6)
    n = 1 << k
    raw = C.sample_points(7000 + seed, n + 2)
    pts = [O.g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range(n + 2)]
    pk = I.IpaProvingKey(k, pts[:n], pts[n], pts[n + 1])
    npoly = rng.randrange(1, 7)
    polys = [[rnd() for _ in range(n)] for _ in range(npoly)]
    blinds = [rnd() for _ in polys]
    coms = [pk.commit(p, b) for p, b in zip(polys, blinds)]
    w = pow(5, (O.R - 1) // n, O.R)
    shifts = [pow(w, e % n, O.R) for e in (0, 1, -1, 2, -3)][: min(5, n)]
    x = rnd()
    spec = [(rng.randrange(npoly), rng.randrange(len(shifts))) for _ in range(rng.randrange(1, 12))]
    spec += [(p, 0) for p in range(npoly) if all(q != p for q, _ in spec)][:2]
    queries = [(p, shifts[s], I.poly_eval(polys[p], x * shifts[s] % O.R)) for p, s in spec]
    kind = rng.choice(["evm", "poseidon"])
    tk, Tr = TR[kind]
    t = Tr()
    I.bgh19_create_proof(pk, polys, blinds, x, queries, t, rnd)
    proof = t.finalize()
    exp = I.bgh19_verify(pk.g[0], pk.h, pk.s, [K.Msm.base(c) for c in coms], x, queries,
                         I.bgh19_read_proof(k, queries, Tr(proof)))
    assert I.ipa_decide(pk.g, exp)
    out = _buf(32 * k + 64)
    cm = pack_commitments([K.Msm.base(p) for p in coms])
    assert H.hd_ipa_bgh19_verify(tk, pack_svk(k, pk.g[0], pk.h, pk.s), cm, O.fe_to_bytes(x), pack_queries(queries), proof,
                                 len(proof), out) == 1
    assert out.raw == pack_acc(exp)
    assert H.hd_ipa_decide_all(k, raw[:64 * n], n, out.raw, 1) == 1


def _batch_args(H_fn):
    H_fn.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                     ctypes.c_size_t, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint,
                     ctypes.c_char_p, ctypes.c_int]


@pytest.mark.parametrize("kind", ["evm", "poseidon"])
def test_plonk_over_ipa_batch_is_one_launch_and_matches_single(H, kind):
    """N proofs of one protocol: `plonk_ipa_verify_batch` (all 2N succinct-check MSMs in one segmented
    launch, then `decide_all` with four accumulators in flight) returns, proof by proof, the accumulators
    of the one-at-a-time path = the oracle's; one bad proof anywhere fails the batch."""
    import struct

    import plonk as P
    import plonk_synth as S

    k, n = 5, 12
    P_, S_, rng, pr, dl, inst0, kd, g, gb, h, s = _plonk_ipa_setup("batch-%s" % kind, k, num_instance=(2,))
    tk, Tr = TR[kind]
    insts = [[[rng.randrange(O.R) for _ in range(2)]] for _ in range(n)]
    proofs = [P.forge_proof_ipa(pr, insts[i], kd, Tr, rng, dl) for i in range(n)]
    want = b"".join(pack_acc(P.succinct_verify_ipa(g[0], h, s, pr, insts[i], P.plonk_proof_read(pr, insts[i], Tr(proofs[i]), "bgh19"))[0])
                    for i in range(n))
    _batch_args(H.hd_plonk_ipa_verify_batch)
    pb = S.pack_protocol(pr)
    ib = b"".join(S.pack_instances(x) for x in insts)
    prb = b"".join(struct.pack("<I", len(p)) + p for p in proofs)
    svk = pack_svk(k, g[0], h, s)
    out = _buf((32 * k + 64) * n)
    for threads in (1, 4):
        assert H.hd_plonk_ipa_verify_batch(tk, pb, len(pb), ib, len(ib), prb, len(prb), n, svk, gb, len(gb) // 64, threads, out, 1) == 1
        assert out.raw == want
    bad = list(proofs)
    bad[7] = bad[7][:40] + bytes([bad[7][40] ^ 1]) + bad[7][41:]
    prb2 = b"".join(struct.pack("<I", len(p)) + p for p in bad)
    assert H.hd_plonk_ipa_verify_batch(tk, pb, len(pb), ib, len(ib), prb2, len(prb2), n, svk, gb, len(gb) // 64, 2, out, 1) in (0, -10)
    # a wrong committing key: every succinct check passes, decide_all rejects
    gb2 = gb[:64] + gb[128:192] + gb[128:]
    assert H.hd_plonk_ipa_verify_batch(tk, pb, len(pb), ib, len(ib), prb, len(prb), n, svk, gb2, len(gb2) // 64, 2, out, 0) == 1
    assert H.hd_plonk_ipa_verify_batch(tk, pb, len(pb), ib, len(ib), prb, len(prb), n, svk, gb2, len(gb2) // 64, 2, out, 1) == 0


@pytest.mark.parametrize("k,world", [(1, 2), (6, 3), (12, 8), (16, 5)])
def test_sharded_ipa_commit_emulated_ranks(gpu_ctx, k, world):
    """Multi-GPU `IpaAs::decide` emulated on one device: `world` shards of the committing key
    (`snarkv_ipa_dk_create_shard`) each commit to their slice (`snarkv_ipa_commit_partial_dev`), the folded
    partials equal the point the un-sharded decider accepts; a shard key cannot decide on its own."""
    import torch

    import snark_verifier_amd as sv
    from snark_verifier_amd.distributed import shard_range

    rnd = random.Random(k * 10 + world)
    n = 1 << k
    gb = C.sample_points(2000 + k, n)
    xi = [rnd.randrange(O.R) for _ in range(k)]
    xb = b"".join(O.fe_to_bytes(x) for x in xi)
    parts = torch.zeros(world, sv.G1_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    keys = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        if hi > lo:
            dk = sv.IpaDecidingKey(gpu_ctx, gb[64 * lo:64 * hi], k, lo)
            keys.append(dk)
            gpu_ctx.ipa_commit_partial_dev(dk, xb, parts[r].data_ptr())
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    gpu_ctx.fold_partials_dev(parts.data_ptr(), world, out.data_ptr())
    gpu_ctx.sync()
    u = bytes(out.cpu().numpy())
    full = sv.IpaDecidingKey(gpu_ctx, gb)
    assert gpu_ctx.ipa_decide_batch(full, xb, u) == [True]
    assert u == C.msm_pippenger(b"".join(O.fe_to_bytes(c) for c in I.h_coeffs(xi, 1)), gb, 2)
    with pytest.raises(sv.SnarkvError):
        gpu_ctx.ipa_decide_batch(keys[0], xb, u)
    with pytest.raises(sv.SnarkvError):
        sv.IpaDecidingKey(gpu_ctx, gb, k, 1)  # shard past the end of the key
