"""CPU: the DEVICE math headers (csrc/fq.h, g1.h, tower.h, pairing.h)
compiled for the host and checked against the big-integer oracle -- catches
formula bugs before any GPU minute is spent."""
import ctypes
import random

import bn254 as O


def _buf(n):
    return ctypes.create_string_buffer(n)


def test_fq_ops(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(7)
    vals = [0, 1, 2, O.P - 1, O.P - 2] + [rng.randrange(O.P) for _ in range(60)]
    o = _buf(32)
    for a in vals:
        for b in vals[:12]:
            L.ht_fq_mul(O.fe_to_bytes(a), O.fe_to_bytes(b), o)
            assert O.fe_from_bytes(o.raw) == a * b % O.P
            L.ht_fq_add(O.fe_to_bytes(a), O.fe_to_bytes(b), o)
            assert O.fe_from_bytes(o.raw) == (a + b) % O.P
            L.ht_fq_sub(O.fe_to_bytes(a), O.fe_to_bytes(b), o)
            assert O.fe_from_bytes(o.raw) == (a - b) % O.P
    a = rng.randrange(1, O.P)
    L.ht_fq_inv(O.fe_to_bytes(a), o)
    assert O.fe_from_bytes(o.raw) == pow(a, -1, O.P)


def test_g1_group_law_and_exceptional_cases(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(8)
    Pt = O.g1_mul(O.G1_GEN, rng.randrange(O.R))
    Q = O.g1_mul(O.G1_GEN, rng.randrange(O.R))
    o = _buf(64)
    for x, y in [(Pt, Q), (Pt, Pt), (Pt, O.g1_neg(Pt)), (None, Q), (Pt, None), (None, None)]:
        L.ht_g1_add(O.g1_to_bytes(x), O.g1_to_bytes(y), o)
        assert O.g1_from_bytes(o.raw) == O.g1_add(x, y)
    for k in [0, 1, 2, 3, O.R - 1, O.R, rng.randrange(O.R), (1 << 256) - 1]:
        L.ht_g1_mul(O.g1_to_bytes(Pt), int(k).to_bytes(32, "little"), o)
        assert O.g1_from_bytes(o.raw) == O.g1_mul(Pt, k)
    for k1, k2, A, B in [(5, 7, Pt, Q), (5, 5, Pt, Pt), (5, O.R - 5, Pt, Pt), (0, 3, Pt, Q), (3, 0, Pt, Q)]:
        L.ht_g1_lincomb(O.g1_to_bytes(A), k1.to_bytes(32, "little"), O.g1_to_bytes(B), k2.to_bytes(32, "little"), o)
        assert O.g1_from_bytes(o.raw) == O.g1_add(O.g1_mul(A, k1), O.g1_mul(B, k2))
    assert L.ht_g1_on_curve(O.g1_to_bytes(Pt)) == 1
    assert L.ht_g1_on_curve(O.g1_to_bytes((Pt[0], (Pt[1] + 1) % O.P))) == 0


def _rfq12(rng):
    f2 = lambda: O.Fq2(rng.randrange(O.P), rng.randrange(O.P))  # noqa: E731
    return O.Fq12(O.Fq6(f2(), f2(), f2()), O.Fq6(f2(), f2(), f2()))


def test_fq12_tower(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(9)
    A, B = _rfq12(rng), _rfq12(rng)
    o = _buf(384)
    L.ht_fq12_mul(A.to_bytes(), B.to_bytes(), o)
    assert o.raw == (A * B).to_bytes()
    L.ht_fq12_sqr(A.to_bytes(), o)
    assert o.raw == (A * A).to_bytes()
    L.ht_fq12_inv(A.to_bytes(), o)
    assert o.raw == A.inv().to_bytes()
    for k in (1, 2, 3):
        L.ht_fq12_frob(A.to_bytes(), k, o)
        assert o.raw == A.pow(O.P ** k).to_bytes()


def test_final_exponentiation_is_exact(hosttest_lib):
    rng = random.Random(10)
    A = _rfq12(rng)
    o = _buf(384)
    hosttest_lib.ht_final_exp(A.to_bytes(), o)
    assert o.raw == O.final_exponentiation(A).to_bytes()


def test_pairing_matches_oracle_value(hosttest_lib, golden_decider):
    L = hosttest_lib
    rng = random.Random(11)
    a, b = rng.randrange(O.R), rng.randrange(O.R)
    Pa, Qb = O.g1_mul(O.G1_GEN, a), O.g2_mul(O.G2_GEN, b)
    o = _buf(384)
    L.ht_pairing_product(O.g1_to_bytes(Pa), O.g2_to_bytes(Qb), 1, o)
    assert o.raw == O.pairing(Pa, Qb).to_bytes()
    g2 = bytes.fromhex(golden_decider["g2"])
    ns_g2 = O.g2_to_bytes(O.g2_neg(O.g2_from_bytes(bytes.fromhex(golden_decider["s_g2"]))))
    for case in golden_decider["cases"]:
        acc = bytes.fromhex(case["acc"])
        L.ht_pairing_product(acc, g2 + ns_g2, 2, o)
        assert o.raw == bytes.fromhex(case["gt"]), case["name"]
        assert (o.raw == O.FQ12_ONE.to_bytes()) == case["accept"]


# ---------------------------------------------------------------------------
# 9x29-bit lazy field + group law (csrc/fq29.h, g1_29.h): the MSM hot path
# ---------------------------------------------------------------------------
def _rand_point(rng):
    while True:
        x = rng.randrange(O.P)
        r = (x ** 3 + 3) % O.P
        y = pow(r, (O.P + 1) // 4, O.P)
        if y * y % O.P == r:
            return (x, y)


def test_fq29_mul_sqr_lazy(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(5)
    fb = O.fe_to_bytes
    vals = [0, 1, 2, O.P - 1, O.P - 2, 1 << 253, (1 << 253) + 12345] + [rng.randrange(O.P) for _ in range(120)]
    o = _buf(32)
    for a in vals:
        for b in vals[:10] + [rng.choice(vals)]:
            L.ht29_fq_mul(fb(a), fb(b), o)
            assert O.fe_from_bytes(o.raw) == a * b % O.P
        L.ht29_fq_sqr(fb(a), o)
        assert O.fe_from_bytes(o.raw) == a * a % O.P
    for _ in range(200):
        a, b, c, d, e = [rng.choice(vals) for _ in range(5)]
        L.ht29_fq_lazy_expr(fb(a), fb(b), fb(c), fb(d), fb(e), o)
        assert O.fe_from_bytes(o.raw) == ((a - b) * (c + d) - e) % O.P
        assert L.ht29_is_zero_mod_p_of_diff(fb(a), fb(b)) == (1 if a == b else 0)
        assert L.ht29_is_zero_mod_p_of_diff(fb(a), fb(a)) == 1
    # binary-Euclid inverse (fq29_inv) vs Python, and vs the Fermat chain it replaced
    o2 = _buf(32)
    edge = [1, 2, 3, O.P - 1, O.P - 2, (O.P + 1) // 2, 1 << 253, (1 << 253) - 1, 0x30644e72 << 224]
    for a in edge + [rng.randrange(1, O.P) for _ in range(200)]:
        L.ht29_fq_inv(fb(a), o)
        assert O.fe_from_bytes(o.raw) == pow(a, -1, O.P), hex(a)
    for a in [rng.randrange(1, O.P) for _ in range(3)]:
        L.ht29_fq_inv_fermat(fb(a), o2)
        assert O.fe_from_bytes(o2.raw) == pow(a, -1, O.P)
    # the plain-integer kernels underneath: safegcd (shipped) and binary Euclid agree with pow()
    for a in edge + [rng.randrange(1, O.P) for _ in range(2000)] + [(1 << k) % O.P for k in range(0, 254, 7)] + \
            [O.P - (1 << k) for k in range(0, 253, 11)]:
        L.ht_words_inv_safegcd(a.to_bytes(32, "little"), o)
        assert int.from_bytes(o.raw, "little") == pow(a, -1, O.P), hex(a)
    for a in edge + [rng.randrange(1, O.P) for _ in range(50)]:
        L.ht_words_inv_binary(a.to_bytes(32, "little"), o)
        assert int.from_bytes(o.raw, "little") == pow(a, -1, O.P), hex(a)
    L.ht_words_inv_safegcd(bytes(32), o)
    assert o.raw == bytes(32)
    L.ht29_fq_inv(fb(0), o)  # 0 -> 0, as a^(p-2) gives
    assert O.fe_from_bytes(o.raw) == 0
    for _ in range(20):  # lazy, possibly negative input
        a, b = rng.randrange(O.P), rng.randrange(O.P)
        if a != b:
            L.ht29_fq_inv_of_diff(fb(a), fb(b), o)
            assert O.fe_from_bytes(o.raw) == pow(a - b, -1, O.P)


def test_g1_29_fast_and_careful_adders(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(6)
    o = _buf(64)
    for n in (1, 2, 3, 10, 40):
        pts = [_rand_point(rng) for _ in range(n)]
        sg = bytes(rng.randrange(2) for _ in range(n))
        exp = None
        for p, s in zip(pts, sg):
            exp = O.g1_add(exp, O.g1_neg(p) if s else p)
        raw = b"".join(O.g1_to_bytes(p) for p in pts)
        for careful in (0, 1):
            assert L.ht29_madd_chain(raw, sg, n, careful, o) == 0
            assert O.g1_from_bytes(o.raw) == exp
        tot = None
        for p in pts:
            tot = O.g1_add(tot, p)
        for h in (0, 1, n // 2, n):
            for careful in (0, 1):
                assert L.ht29_add_halves(raw, n, h, careful, o) == 0
                assert O.g1_from_bytes(o.raw) == tot


def test_g1_29_exceptional_cases_are_flagged_then_handled(hosttest_lib):
    """The fast adders must FLAG (degenerate ZZ) whatever they cannot compute;
    the careful adders must compute it (duplicate / opposite bases are legal)."""
    L = hosttest_lib
    rng = random.Random(7)
    p0, p1 = _rand_point(rng), _rand_point(rng)
    s01 = O.g1_add(p0, p1)
    o = _buf(64)
    cases = [
        ([p0, p0], [0, 0]), ([p0, p0], [0, 1]), ([p0, p1, p0], [0, 0, 0]), ([p0, p1, s01], [0, 0, 1]),
        ([p0, None, p1], [0, 0, 0]), ([None, None], [0, 0]), ([p0, p1, s01, p1], [0, 0, 1, 0]),
        ([p0, p1, s01], [0, 0, 0]),
    ]
    for pts, sg in cases:
        exp = None
        for p, s in zip(pts, sg):
            exp = O.g1_add(exp, O.g1_neg(p) if s else p)
        raw = b"".join(O.g1_to_bytes(p) for p in pts)
        L.ht29_madd_chain(raw, bytes(sg), len(pts), 1, o)
        assert O.g1_from_bytes(o.raw) == exp
        if None not in pts:
            flagged = L.ht29_madd_chain(raw, bytes(sg), len(pts), 0, o)
            assert flagged == 1 or O.g1_from_bytes(o.raw) == exp
    raw = O.g1_to_bytes(p0) + O.g1_to_bytes(p0)
    assert L.ht29_add_halves(raw, 2, 1, 1, o) == 0 and O.g1_from_bytes(o.raw) == O.g1_double(p0)
    assert L.ht29_add_halves(raw, 2, 1, 0, o) == 1
    raw = O.g1_to_bytes(p0) + O.g1_to_bytes(O.g1_neg(p0))
    L.ht29_add_halves(raw, 2, 1, 1, o)
    assert O.g1_from_bytes(o.raw) is None
    assert L.ht29_add_halves(raw, 2, 1, 0, o) == 1


def test_g1_29_scalar_mul(hosttest_lib):
    L = hosttest_lib
    rng = random.Random(8)
    p0 = _rand_point(rng)
    o = _buf(64)
    for k in [0, 1, 2, 3, 5, O.R - 1, rng.randrange(O.R), rng.randrange(O.R)]:
        for careful in (0, 1):
            L.ht29_g1_mul(O.g1_to_bytes(p0), int(k).to_bytes(32, "little"), careful, o)
            assert O.g1_from_bytes(o.raw) == O.g1_mul(p0, k)
    for k in [O.R, O.R + 1, O.R + 2, (1 << 256) - 1]:  # non-canonical scalars hit P = +-Q: careful path
        L.ht29_g1_mul(O.g1_to_bytes(p0), int(k).to_bytes(32, "little"), 1, o)
        assert O.g1_from_bytes(o.raw) == O.g1_mul(p0, k)


def test_g1_29_jacobian_double_chain(hosttest_lib):
    P = O.g1_mul(O.G1_GEN, 987654321)
    o = _buf(64)
    for n in (0, 1, 2, 16, 112, 240):
        hosttest_lib.ht29_double_n(O.g1_to_bytes(P), n, o)
        assert O.g1_from_bytes(o.raw) == O.g1_mul(P, pow(2, n + 1, O.R))


def test_fq29_fused_two_product(hosttest_lib):
    """fq29_mul2: a*b + c*d (and a*b - c*d through a limb-wise negated c) with one reduction."""
    rng = random.Random(21)
    o = _buf(32)
    fb = O.fe_to_bytes
    vals = [0, 1, O.P - 1, O.P - 2] + [rng.randrange(O.P) for _ in range(6)]
    for a in vals:
        for c in vals[:6]:
            b, d = rng.randrange(O.P), O.P - 1
            for neg in (0, 1):
                hosttest_lib.ht29_fq_mul2(fb(a), fb(b), fb(c), fb(d), neg, o)
                exp = (a * b + (-c if neg else c) * d) % O.P
                assert O.fe_from_bytes(o.raw) == exp


def test_cooperative_fq12_coop3_rounds(hosttest_lib):
    """The round the shipped k_decide runs (pairing_coop29.h coop3: 72 fused
    two-product lanes, low/high sums, xi fix-up in the finalize step), emulated
    lane by lane: exact over long chains of products and squarings, including
    all-(p-1) coefficients (worst-case magnitudes)."""
    rng = random.Random(14)
    o = _buf(384)
    for rounds, mode in ((1, 0), (2, 0), (7, 0), (40, 0), (1, 1), (3, 1), (25, 1)):
        A, B = _rfq12(rng), _rfq12(rng)
        hosttest_lib.ht_coop3_fq12_mul_iter(A.to_bytes(), B.to_bytes(), rounds, mode, o)
        exp = A
        for _ in range(rounds):
            if mode == 1:
                exp = exp * exp
            exp = exp * B
        assert o.raw == exp.to_bytes(), (rounds, mode)
    m = O.Fq2(O.P - 1, O.P - 1)
    M = O.Fq12(O.Fq6(m, m, m), O.Fq6(m, m, m))
    for mode in (0, 1):
        hosttest_lib.ht_coop3_fq12_mul_iter(M.to_bytes(), M.to_bytes(), 20, mode, o)
        exp = M
        for _ in range(20):
            if mode == 1:
                exp = exp * exp
            exp = exp * M
        assert o.raw == exp.to_bytes()
    # sparse line-shaped multiplier (only w^0, w^1, w^3 non-zero), as in the Miller loop
    z = O.Fq2(0, 0)
    for _ in range(3):
        A = _rfq12(rng)
        c = [O.Fq2(rng.randrange(O.P), rng.randrange(O.P)) for _ in range(3)]
        # tower: c0 = (w^0, w^2, w^4), c1 = (w^1, w^3, w^5)
        Lm = O.Fq12(O.Fq6(c[0], z, z), O.Fq6(c[1], c[2], z))
        hosttest_lib.ht_coop3_fq12_mul_iter(A.to_bytes(), Lm.to_bytes(), 3, 1, o)
        exp = A
        for _ in range(3):
            exp = exp * exp * Lm
        assert o.raw == exp.to_bytes()


def test_fr29_scalar_field(hosttest_lib):
    """fr29.h (the Poseidon kernel's field): Montgomery product, lazy add, x^5, codecs, vs Python mod r."""
    rng = random.Random(31)
    o = _buf(32)
    fb = lambda x: (x % O.R).to_bytes(32, "little")
    vals = [0, 1, 2, O.R - 1, O.R - 2, (O.R - 1) // 2] + [rng.randrange(O.R) for _ in range(40)]
    for a in vals:
        hosttest_lib.ht_fr29_roundtrip(fb(a), o)
        assert int.from_bytes(o.raw, "little") == a
        b, c = rng.choice(vals), rng.choice(vals)
        hosttest_lib.ht_fr29_expr(fb(a), fb(b), fb(c), o)
        assert int.from_bytes(o.raw, "little") == pow((a * b + c) % O.R, 5, O.R)


def test_packed_point_memory_form_roundtrip(hosttest_lib):
    """G1Packed (g1_29.h): the 64-byte memory form of the Montgomery points k_accumulate gathers -- each coordinate's
    canonical Montgomery residue x * 2^261 mod p as a 256-bit LE integer -- unpacks to exactly the 9 x 29-bit limbs it
    was packed from, for random points, the identity and coordinates near 0 and p."""
    import random

    H = hosttest_lib
    H.ht_g1_pack_roundtrip.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    rng = random.Random(64)
    R261 = pow(2, 261, O.P)
    pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(40)]
    raw = [O.g1_to_bytes(p) for p in pts] + [bytes(64)]
    # field values that are not curve points exercise the codec alone: 0 / 1 / p-1 / all 29-bit-limb boundaries
    for x in (1, 2, O.P - 1, O.P - 2, (1 << 253) + 12345, (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232):
        raw.append(O.fe_to_bytes(x % O.P) + O.fe_to_bytes((O.P - x) % O.P))
    for b in raw:
        out, packed = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        assert H.ht_g1_pack_roundtrip(b, out, packed) == 1
        assert out.raw == b
        if b != bytes(64):
            x, y = int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little")
            # the packed form (g1_29.h): each coordinate's canonical Montgomery residue as a plain 256-bit integer
            for raw32, v in ((packed.raw[:32], x * R261 % O.P), (packed.raw[32:], y * R261 % O.P)):
                assert raw32 == v.to_bytes(32, "little")
        else:
            assert packed.raw == bytes(64)


def test_program_driven_decide_kernel_on_the_host(hosttest_lib, golden_decider):
    """decide_w.h + decide_sched.hpp: the whole scheduled program of the latency decide kernel (Miller loop with the line
    products computed ahead, exact final exponentiation) run lane by lane on the host gives the oracle's Gt bytes on every
    golden accumulator (accepting, rejecting, identity pairs), the schedule never writes a register its round reads, and
    it fits the LDS registers."""
    import ctypes

    L = hosttest_lib
    g2 = bytes.fromhex(golden_decider["g2"])
    ns_g2 = O.g2_to_bytes(O.g2_neg(O.g2_from_bytes(bytes.fromhex(golden_decider["s_g2"]))))
    info = (ctypes.c_int * 4)()
    o = _buf(384)
    for case in golden_decider["cases"]:
        assert L.ht_decide_w(g2 + ns_g2, bytes.fromhex(case["acc"]), o, info) == 0, case["name"]
        assert o.raw == bytes.fromhex(case["gt"]), case["name"]
    rounds, regs, crit, nops = list(info)
    assert crit <= rounds <= crit + 40, (rounds, crit)  # the two duos keep the critical path fed
    assert regs <= 16 and nops > 500


def test_montgomery_boundary_codecs(hosttest_lib):
    """SNARKV_FLAG_MONTGOMERY: the device's codecs for halo2curves' in-memory form (a * 2^256 mod p in 4 x u64) against plain
    integer arithmetic -- Fq both ways, the Fr scalar reduction, whole points incl. the identity."""
    import mont_util as M

    L = hosttest_lib
    rng = random.Random(77)
    o = _buf(32)
    for v in [0, 1, 2, O.P - 1, O.P - 2, (1 << 253) + 5, (1 << 29) - 1, 1 << 232] + [rng.randrange(O.P) for _ in range(40)]:
        b = v.to_bytes(32, "little")
        L.ht_mont_codec(b, 0, o)
        assert o.raw == M.fe_to_mont(b, O.P), v
        L.ht_mont_codec(b, 1, o)
        assert o.raw == M.fe_from_mont(b, O.P), v
    for v in [0, 1, O.R - 1, (1 << 253) + 9] + [rng.randrange(O.R) for _ in range(40)]:
        b = v.to_bytes(32, "little")
        L.ht_mont_codec(M.fe_to_mont(b, O.R), 2, o)
        assert o.raw == b, v
    o64 = _buf(64)
    for pt in [O.g1_to_bytes(O.g1_mul(O.G1_GEN, rng.randrange(1, O.R))) for _ in range(8)] + [bytes(64)]:
        L.ht_g1_recode(pt, 0, 1, o64)
        assert o64.raw == M.coords_to_mont(pt)
        L.ht_g1_recode(M.coords_to_mont(pt), 1, 0, o64)
        assert o64.raw == pt
        L.ht_g1_recode(M.coords_to_mont(pt), 1, 1, o64)
        assert o64.raw == M.coords_to_mont(pt)


def test_wavefront_parallel_g2_line_tables_on_the_host(hosttest_lib, golden_decider):
    """g2_prepare_w.h + the generated level program: every line coefficient of both points of the golden key (g2, -s_g2) and of
    random G2 points equals pairing.h's one-lane g2_prepare in the 29-bit canonical form the decide kernels read."""
    L = hosttest_lib
    g2, s_g2 = bytes.fromhex(golden_decider["g2"]), bytes.fromhex(golden_decider["s_g2"])
    assert L.ht_g2_prepare_w(g2, 0) == 0
    assert L.ht_g2_prepare_w(s_g2, 1) == 0
    rng = random.Random(31)
    for _ in range(3):
        q = O.g2_to_bytes(O.g2_mul(O.G2_GEN, rng.randrange(1, O.R)))
        assert L.ht_g2_prepare_w(q, 0) == 0
    assert L.ht_g2_prepare_w(bytes(128), 0) == -1
