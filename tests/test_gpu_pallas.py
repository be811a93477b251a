"""GPU: the pasta build (libsnarkv_pallas.so) -- the same Pippenger and IPA-decider kernels compiled
for pallas -- against the pallas oracle: MSM bytes, and the reference's `test_ipa` / `test_ipa_as`
(pcs/ipa.rs:434-466, pcs/ipa/accumulation.rs:240-290) with the device doing `IpaAs::decide`."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import bn254 as BN  # noqa: E402
import ipa as I  # noqa: E402
import pallas as PA  # noqa: E402
import transcript as T  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pctx():
    from snark_verifier_amd import pallas as PL

    c = PL.PallasContext(0)
    yield c
    c.close()


@pytest.fixture()
def on_pallas():
    I.use_curve(PA)
    yield
    I.use_curve(BN)


def _pack(scalars, points):
    return b"".join(PA.fe_to_bytes(s) for s in scalars), b"".join(PA.g1_to_bytes(p) for p in points)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 257, 1500])
def test_msm_against_the_pallas_oracle(pctx, n):
    rnd = random.Random(n)
    pts = PA.sample_points(n, n)
    sc = [rnd.randrange(PA.R) for _ in range(n)]
    s, p = _pack(sc, pts)
    assert pctx.msm_pippenger(s, p) == PA.g1_to_bytes(PA.g1_msm_pippenger(sc, pts))


def test_msm_edge_cases(pctx):
    import snark_verifier_amd as sv

    rnd = random.Random(1)
    base = PA.sample_points(5, 40)
    # duplicates, P and -P, zero / r-1 / tiny scalars, identity points: the exceptional cases of the adders
    pts = base[:10] + base[:10] + [PA.g1_neg(q) for q in base[:10]] + [None] * 3 + base[10:]
    sc = [rnd.randrange(PA.R) for _ in pts]
    for i, v in ((0, 0), (1, PA.R - 1), (2, 1), (3, 2), (11, sc[1]), (21, sc[1])):
        sc[i] = v
    s, p = _pack(sc, pts)
    want = PA.g1_msm_pippenger([x for x, q in zip(sc, pts) if q is not None], [q for q in pts if q is not None])
    assert pctx.msm_pippenger(s, p) == PA.g1_to_bytes(want)
    # everything cancels -> the identity, 64 zero bytes
    s2, p2 = _pack([7, 7], [base[0], PA.g1_neg(base[0])])
    assert pctx.msm_pippenger(s2, p2) == bytes(64)
    # all scalars equal (one bucket per window holds everything)
    s3, p3 = _pack([12345] * 300, base[:30] * 10)
    assert pctx.msm_pippenger(s3, p3) == PA.g1_to_bytes(PA.g1_mul(PA.g1_msm_pippenger([10] * 30, base[:30]), 12345))
    with pytest.raises(sv.SnarkvError):
        pctx.msm_pippenger(b"", b"")
    with pytest.raises(sv.SnarkvError):
        pctx.msm_pippenger(s2, p2[:64])


def test_msm_2p16_linearity_and_prefix(pctx):
    """2^16 terms (1 024 distinct points, each 64 times): the pure-Python oracle is too slow for that
    many, so size-independent properties -- MSM(s, P) + MSM(t, P) = MSM(s + t, P) -- and the same sum
    folded on the host to a 1 024-term oracle MSM."""
    n = 1 << 16
    rnd = random.Random(3)
    base = PA.sample_points(11, 1 << 10)
    sa = [rnd.randrange(PA.R) for _ in range(n)]
    sb = [rnd.randrange(PA.R) for _ in range(n)]
    pb = b"".join(PA.g1_to_bytes(p) for p in base) * (n >> 10)
    enc = lambda v: b"".join(PA.fe_to_bytes(x) for x in v)  # noqa: E731
    a = PA.g1_from_bytes(pctx.msm_pippenger(enc(sa), pb))
    b = PA.g1_from_bytes(pctx.msm_pippenger(enc(sb), pb))
    ab = PA.g1_from_bytes(pctx.msm_pippenger(enc([(x + y) % PA.R for x, y in zip(sa, sb)]), pb))
    assert PA.g1_add(a, b) == ab and ab is not None
    # collapse the repeats on the host: sum_j s_{i + 1024 j} per base point, 1024-term oracle MSM
    folded = [sum(sa[i::1 << 10]) % PA.R for i in range(1 << 10)]
    assert a == PA.g1_msm_pippenger(folded, base)


@pytest.mark.parametrize("k,zk", [(6, False), (6, True), (10, True)])
def test_ipa_decide_on_pallas(pctx, on_pallas, k, zk):
    rnd = random.Random("gpu-pallas-%d-%d" % (k, zk))
    rng = lambda: rnd.randrange(PA.R)  # noqa: E731
    n = 1 << k
    pts = PA.sample_points(100 + k + zk, n + 2)
    pk = I.IpaProvingKey(k, pts[:n], pts[n], pts[n + 1] if zk else None)
    dk = pctx.ipa_dk_create(b"".join(PA.g1_to_bytes(p) for p in pk.g))
    assert dk.k == k
    accs = []
    for _ in range(2 if k == 10 else 3):
        p = [rng() for _ in range(n)]
        omega, z = (rng() if zk else None), rng()
        c = pk.commit(p, omega)
        t = T.Blake2bTranscript(PA)
        I.ipa_create_proof(pk, p, z, omega, t, rng)
        accs.append(I.ipa_succinct_verify(pk.h, pk.s, [(1, c)], z, I.poly_eval(p, z),
                                          I.ipa_read_proof(zk, k, T.Blake2bTranscript(PA, t.finalize()))))
    t = T.Blake2bTranscript(PA)
    I.ipa_as_create_proof(pk, accs, t, rng)
    new = I.ipa_as_verify(pk.h, pk.s, accs, I.ipa_as_read_proof(zk, k, accs, T.Blake2bTranscript(PA, t.finalize())))
    everything = accs + [new]
    xi = b"".join(PA.fe_to_bytes(x) for a in everything for x in a[0])
    u = b"".join(PA.g1_to_bytes(a[1]) for a in everything)
    assert pctx.ipa_decide_batch(dk, xi, u) == [True] * len(everything)
    bad_u = PA.g1_to_bytes(PA.g1_add(new[1], pk.h))
    bad_xi = b"".join(PA.fe_to_bytes(x) for x in [(new[0][0] + 1) % PA.R] + new[0][1:])
    good_xi = b"".join(PA.fe_to_bytes(x) for x in new[0])
    assert pctx.ipa_decide_batch(dk, good_xi + bad_xi + good_xi, bad_u + PA.g1_to_bytes(new[1]) * 2) == [False, False, True]
    dk.close()


def test_both_libraries_in_one_process(pctx, gpu_ctx):
    """The BN254 library and the pasta build side by side: same kernels, different constants and
    namespaces (-Dsnarkv=snarkv_pallas, -Bsymbolic) -- each answers for its own curve."""
    import coracle as C

    rnd = random.Random(8)
    n = 200
    s = b"".join(rnd.randrange(BN.R).to_bytes(32, "little") for _ in range(n))
    pb = C.sample_points(3, n)
    assert gpu_ctx.msm_pippenger(s, pb) == C.msm_pippenger(s, pb, 1)
    pts = PA.sample_points(4, n)
    sc = [rnd.randrange(PA.R) for _ in range(n)]
    s2, p2 = _pack(sc, pts)
    assert pctx.msm_pippenger(s2, p2) == PA.g1_to_bytes(PA.g1_msm_pippenger(sc, pts))
    assert gpu_ctx.msm_pippenger(s, pb) == C.msm_pippenger(s, pb, 1)


def test_native_loader_msm_on_pallas(pctx):
    """`NativeLoader::multi_scalar_multiplication` semantics on pallas (loader/native.rs:61-71): single
    and segmented small MSMs (every chunking the kernels pick by batch size) against the oracle's naive
    sum; empty segment / validation errors."""
    import snark_verifier_amd as sv

    rnd = random.Random(12)
    pts = PA.sample_points(6, 64)
    for n in (1, 2, 3, 24, 64):
        sc = [rnd.randrange(PA.R) for _ in range(n)]
        sc[0] = rnd.choice([0, 1, PA.R - 1, sc[0]])
        s, p = _pack(sc, pts[:n])
        assert pctx.msm_naive(s, p) == PA.g1_to_bytes(PA.g1_msm_naive(sc, pts[:n]))
    # 40 segments of ragged sizes, duplicates and opposite points inside a segment
    offs, sc_all, pt_all, want = [0], [], [], []
    for j in range(40):
        n = rnd.randrange(1, 30)
        q = [rnd.choice(pts) for _ in range(n)]
        if n > 2:
            q[1] = q[0]
            q[2] = PA.g1_neg(q[0])
        sc = [rnd.randrange(PA.R) for _ in range(n)]
        want.append(PA.g1_msm_naive(sc, q))
        sc_all += sc
        pt_all += q
        offs.append(offs[-1] + n)
    s, p = _pack(sc_all, pt_all)
    assert pctx.msm_batched(s, p, offs) == b"".join(PA.g1_to_bytes(w) for w in want)
    # a big batch: > 16 k terms takes the one-lane-per-(term, half) kernels
    big_pts = (pts * 300)[:18000]
    big_sc = [rnd.randrange(PA.R) for _ in big_pts]
    offs2 = list(range(0, 18001, 1500))
    s, p = _pack(big_sc, big_pts)
    got = pctx.msm_batched(s, p, offs2)
    for j in (0, 5, 11):
        assert got[64 * j:64 * j + 64] == PA.g1_to_bytes(PA.g1_msm_pippenger(big_sc[offs2[j]:offs2[j + 1]], big_pts[offs2[j]:offs2[j + 1]]))
    with pytest.raises(sv.SnarkvError):
        pctx.msm_batched(s[:64], p[:128], [0, 1, 1, 2])  # empty segment: the reference panics (native.rs:69)
    off_curve = PA.fe_to_bytes(5) + PA.fe_to_bytes(7)
    with pytest.raises(sv.SnarkvError):
        pctx.msm_naive(PA.fe_to_bytes(3), off_curve, flags=sv.SNARKV_FLAG_VALIDATE)
    with pytest.raises(sv.SnarkvError):
        pctx.msm_naive(PA.fe_to_bytes(PA.R), PA.g1_to_bytes(pts[0]), flags=sv.SNARKV_FLAG_VALIDATE)  # scalar >= r


def test_ipa_succinct_check_msms_on_the_device(pctx, on_pallas):
    """The two `Msm::evaluate(None)` of `Ipa::succinct_verify` (pcs/ipa.rs:172,177) as ONE segmented device
    launch on pallas, fed by the oracle's transcript + scalar algebra: C_k == c[U] + v'[H']."""
    rnd = random.Random(44)
    rng = lambda: rnd.randrange(PA.R)  # noqa: E731
    k = 5
    pts = PA.sample_points(303, (1 << k) + 2)
    pk = I.IpaProvingKey(k, pts[:1 << k], pts[1 << k], pts[(1 << k) + 1])
    p = [rng() for _ in range(1 << k)]
    omega, z = rng(), rng()
    c = pk.commit(p, omega)
    t = T.Blake2bTranscript(PA)
    I.ipa_create_proof(pk, p, z, omega, t, rng)
    pr = I.ipa_read_proof(True, k, T.Blake2bTranscript(PA, t.finalize()))
    xi = [r[2] for r in pr["rounds"]]
    c_bar, alpha = pr["c_bar_alpha"]
    ev = I.poly_eval(p, z)
    lhs = [(1, c), (alpha, c_bar), ((-pr["omega_prime"]) % PA.R, pk.s), (pr["xi_0"] * ev % PA.R, pk.h)]
    for (l, r, x) in pr["rounds"]:
        lhs += [(pow(x, -1, PA.R), l), (x, r)]
    v_prime = I.h_eval(xi, z) * pr["c"] % PA.R
    rhs = [(pr["c"], pr["u"]), (pr["xi_0"] * v_prime % PA.R, pk.h)]
    s, pp = _pack([a for a, _ in lhs + rhs], [b for _, b in lhs + rhs])
    out = pctx.msm_batched(s, pp, [0, len(lhs), len(lhs) + len(rhs)])
    assert out[:64] == out[64:] != bytes(64)
    # a wrong evaluation breaks the equality
    lhs[3] = (pr["xi_0"] * (ev + 1) % PA.R, pk.h)
    s, pp = _pack([a for a, _ in lhs + rhs], [b for _, b in lhs + rhs])
    out = pctx.msm_batched(s, pp, [0, len(lhs), len(lhs) + len(rhs)])
    assert out[:64] != out[64:]


def test_stream_ordering_primitives_on_the_pasta_library(pctx):
    """`snarkv_pallas_ctx_wait_stream` / `snarkv_pallas_stream_wait_ctx`: the device-resident pallas MSM with inputs torch
    has just written (behind a large fill) and the output copied away by torch at once, 60 times without a host sync."""
    import torch

    rng = random.Random(77)
    n = 600
    sets = []
    for k in range(2):
        sc = [rng.randrange(PA.R) for _ in range(n)]
        pts = PA.sample_points(31 + k, n)
        sb, pb = _pack(sc, pts)
        sets.append((torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda(), torch.frombuffer(bytearray(pb), dtype=torch.uint8).cuda(),
                     PA.g1_to_bytes(PA.g1_msm_pippenger(sc, pts))))
    ds = torch.empty(32 * n, dtype=torch.uint8, device="cuda")
    dp = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    ballast = torch.empty(128 << 20, dtype=torch.uint8, device="cuda")
    kept = torch.zeros(60, 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for it in range(60):
        s, p, _ = sets[it % 2]
        ballast.fill_(it)
        ds.copy_(s)
        dp.copy_(p)
        out = torch.full((64,), 0xAB, dtype=torch.uint8, device="cuda")
        pctx.wait_stream()
        pctx.msm_pippenger_dev(ds.data_ptr(), dp.data_ptr(), n, out.data_ptr())
        pctx.stream_wait()
        kept[it] = out
    torch.cuda.synchronize()
    got = bytes(kept.cpu().numpy())
    assert all(got[64 * it:64 * it + 64] == sets[it % 2][2] for it in range(60))
