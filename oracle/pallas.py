"""pallas (pasta) arithmetic oracle -- TEST INFRASTRUCTURE ONLY.

The curve the reference's IPA tests run on (`halo2_curves::pasta::pallas`,
snark-verifier/src/pcs/ipa.rs:443, pcs/ipa/accumulation.rs:249): y^2 = x^3 + 5 over
p = 2^254 + 45560315531419706090280762371685220353, prime group order
r = 2^254 + 45560315531506369815346746415080538113, generator (-1, 2).  Pure-Python big integers from
the definitions; same function names and byte conventions as `oracle/bn254.py` (32-byte little-endian
field elements, x || y points, identity = 64 zero bytes), so `oracle/ipa.py` runs on either module
(`ipa.use_curve`).

PARITY UNPINNED (the crate is not vendored and nothing runs here); pinned by r * G = O, by the group
law checks in tests/test_pallas_oracle.py and by the halo2 convention G = (-1, 2).
"""
P = (1 << 254) + 45560315531419706090280762371685220353
R = (1 << 254) + 45560315531506369815346746415080538113
B1 = 5
TWO_ADICITY, MULT_GEN = 32, 5  # of the scalar field r (halo2curves pasta Fq: S = 32, generator 5)
G1_GEN = (P - 1, 2)

assert P == 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
assert R == 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001


def fe_to_bytes(v):
    return int(v).to_bytes(32, "little")


def fe_from_bytes(b):
    assert len(b) == 32
    return int.from_bytes(b, "little")


def g1_to_bytes(pt):
    if pt is None:
        return b"\x00" * 64
    return fe_to_bytes(pt[0]) + fe_to_bytes(pt[1])


def g1_from_bytes(b):
    assert len(b) == 64
    x, y = fe_from_bytes(b[:32]), fe_from_bytes(b[32:])
    if x == 0 and y == 0:
        return None
    return (x, y)


def g1_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B1) % P == 0


def g1_neg(pt):
    return None if pt is None else (pt[0], (-pt[1]) % P)


def g1_double(pt):
    if pt is None or pt[1] == 0:
        return None
    x, y = pt
    lam = 3 * x * x * pow(2 * y, -1, P) % P
    x3 = (lam * lam - 2 * x) % P
    return (x3, (lam * (x - x3) - y) % P)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        return g1_double(a) if a[1] == b[1] else None
    lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x3 = (lam * lam - a[0] - b[0]) % P
    return (x3, (lam * (a[0] - x3) - a[1]) % P)


# Jacobian internals (speed only: one inversion per scalar multiplication / MSM)
def _jdbl(p):
    x, y, z = p
    if z == 0 or y == 0:
        return (1, 1, 0)
    a, b = x * x % P, y * y % P
    c = b * b % P
    d = 2 * ((x + b) * (x + b) - a - c) % P
    e = 3 * a % P
    x3 = (e * e - 2 * d) % P
    return (x3, (e * (d - x3) - 8 * c) % P, 2 * y * z % P)


def _jadd(p, q):
    if p[2] == 0:
        return q
    if q[2] == 0:
        return p
    z1z1, z2z2 = p[2] * p[2] % P, q[2] * q[2] % P
    u1, u2 = p[0] * z2z2 % P, q[0] * z1z1 % P
    s1, s2 = p[1] * q[2] * z2z2 % P, q[1] * p[2] * z1z1 % P
    if u1 == u2:
        return _jdbl(p) if s1 == s2 else (1, 1, 0)
    h, r = (u2 - u1) % P, (s2 - s1) % P
    hh = h * h % P
    hhh, v = h * hh % P, u1 * hh % P
    x3 = (r * r - hhh - 2 * v) % P
    return (x3, (r * (v - x3) - s1 * hhh) % P, p[2] * q[2] * h % P)


def _to_j(pt):
    return (1, 1, 0) if pt is None else (pt[0], pt[1], 1)


def _from_j(p):
    if p[2] == 0:
        return None
    zi = pow(p[2], -1, P)
    zi2 = zi * zi % P
    return (p[0] * zi2 % P, p[1] * zi2 * zi % P)


def g1_mul(pt, k):
    k %= R
    acc, base = (1, 1, 0), _to_j(pt)
    while k:
        if k & 1:
            acc = _jadd(acc, base)
        base = _jdbl(base)
        k >>= 1
    return _from_j(acc)


def g1_msm_naive(scalars, points):
    """`NativeLoader::multi_scalar_multiplication` semantics (loader/native.rs:61-71)."""
    assert len(scalars) == len(points) and len(scalars) > 0
    acc = (1, 1, 0)
    for s, p in zip(scalars, points):
        acc = _jadd(acc, _to_j(g1_mul(p, s)))
    return _from_j(acc)


def g1_msm_pippenger(scalars, points, c=None):
    """`multi_scalar_multiplication_serial` (util/msm.rs:259-304): unsigned c-bit windows top-down,
    2^c - 1 buckets, running-sum trick; c = ceil(ln n) + 2 as the reference (msm.rs:268) unless given."""
    import math

    n = len(scalars)
    assert n == len(points) and n > 0
    if c is None:
        c = 3 if n < 4 else math.ceil(math.log(n)) + 2
    windows = (256 + c - 1) // c
    js = [_to_j(p) for p in points]
    total = (1, 1, 0)
    for w in reversed(range(windows)):
        for _ in range(c):
            total = _jdbl(total)
        buckets = [(1, 1, 0)] * ((1 << c) - 1)
        for s, p in zip(scalars, js):
            d = ((s % R) >> (w * c)) & ((1 << c) - 1)
            if d:
                buckets[d - 1] = _jadd(buckets[d - 1], p)
        run = (1, 1, 0)
        for b in reversed(buckets):
            run = _jadd(run, b)
            total = _jadd(total, run)
    return _from_j(total)


def fq_sqrt(a):
    """Tonelli-Shanks in F_p (p = 1 mod 2^32).  None if `a` is not a square."""
    a %= P
    if a == 0:
        return 0
    if pow(a, (P - 1) // 2, P) != 1:
        return None
    q, s = P - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (P - 1) // 2, P) != P - 1:
        z += 1
    m, c, t, r = s, pow(z, q, P), pow(a, q, P), pow(a, (q + 1) // 2, P)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % P
            i += 1
        b = pow(c, 1 << (m - i - 1), P)
        m, c = i, b * b % P
        t, r = t * c % P, r * b % P
    return r


def sample_points(seed, n):
    """n deterministic curve points (x from a splitmix-style counter, first x with a square RHS;
    the smaller root).  Prime order: every point generates the group."""
    out, ctr = [], seed * 0x9E3779B97F4A7C15 + 1
    while len(out) < n:
        ctr = (ctr * 6364136223846793005 + 1442695040888963407) % (1 << 256)
        x = ctr % P
        y = fq_sqrt(x * x * x + B1)
        if y is not None:
            out.append((x, min(y, P - y)))
    return out
