"""BN254 (bn256) arithmetic oracle -- TEST INFRASTRUCTURE ONLY.

Pure-Python big-integer restatement, from the mathematical definitions, of the
arithmetic that the reference takes from the un-vendored crate
``halo2curves = "0.6.0"`` (reference `snark-verifier/Cargo.toml:14`, re-exported at
`snark-verifier/src/util/arithmetic.rs:13-18`).  Nothing here is shipped or
timed as product; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.

PARITY UNPINNED: the reference holds no known-answer vectors for BN254 G1 / Gt
values, and cannot be compiled here (no Rust toolchain).
This oracle is pinned instead by (a) public constants (2*G, curve orders, the
EIP-197 G2 generator), (b) algebraic invariants (group law, r*P = O,
bilinearity) and (c) agreement with the independent 4x64-limb Montgomery C
restatement in `oracle/c/`.  Boundary outputs are canonical (affine G1 bytes, a
boolean), so any correct implementation yields identical bytes.

Conventions follow the reference's serialisation:
  * Fq / Fr element  -> 32-byte little-endian canonical (`PrimeField::to_repr`,
    used at `snark-verifier/src/util/msm.rs:264`).
  * G1Affine         -> x || y (64 bytes LE); identity = 64 zero bytes
    (halo2curves represents the affine identity as (0, 0)).
  * Fq2              -> c0 || c1.
"""

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
BN_X = 4965661367192848881  # 0x44e992b44a6909f1
ATE_LOOP = 6 * BN_X + 2
B1 = 3
TWO_ADICITY, MULT_GEN = 28, 7  # of the scalar field r: Fr::S, Fr::MULTIPLICATIVE_GENERATOR (halo2curves bn256)

assert P == 36 * BN_X**4 + 36 * BN_X**3 + 24 * BN_X**2 + 6 * BN_X + 1
assert R == 36 * BN_X**4 + 36 * BN_X**3 + 18 * BN_X**2 + 6 * BN_X + 1


# --------------------------------------------------------------------------
# byte codecs
# --------------------------------------------------------------------------
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


# --------------------------------------------------------------------------
# G1: y^2 = x^3 + 3 over Fq, affine big-int formulas. None = identity.
# --------------------------------------------------------------------------
G1_GEN = (1, 2)


def g1_is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return 0 <= x < P and 0 <= y < P and (y * y - x * x * x - B1) % P == 0


def g1_neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def g1_double(pt):
    if pt is None:
        return None
    x, y = pt
    if y == 0:
        return None
    lam = 3 * x * x * pow(2 * y, -1, P) % P
    x3 = (lam * lam - 2 * x) % P
    return (x3, (lam * (x - x3) - y) % P)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        return g1_double(a)
    lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def g1_mul(pt, k):
    """k * pt by left-to-right double-and-add; k taken mod r is NOT applied
    (callers pass canonical scalars < r, as `Fr` guarantees)."""
    k = int(k)
    if k < 0:
        return g1_mul(g1_neg(pt), -k)
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = g1_double(acc)
        if bit == "1":
            acc = g1_add(acc, pt)
    return acc


def g1_msm_naive(scalars, points):
    """`NativeLoader::multi_scalar_multiplication`
    (reference `snark-verifier/src/loader/native.rs:61-71`): sum of
    base*scalar, folded left to right, then `to_affine()`.  Panics on empty
    input there (`reduce().unwrap()`, :69) -> ValueError here."""
    if len(scalars) == 0:
        raise ValueError("empty MSM (reference panics: native.rs:69)")
    assert len(scalars) == len(points)
    acc = None
    for s, pt in zip(scalars, points):
        acc = g1_add(acc, g1_mul(pt, s))
    return acc


def pippenger_window_size(n):
    """`(n as f64).ln().ceil() as usize + 2` (reference `util/msm.rs:268`)."""
    import math

    return int(math.ceil(math.log(n))) + 2


def g1_msm_pippenger(scalars, points):
    """`util::msm::multi_scalar_multiplication_serial`
    (reference `snark-verifier/src/util/msm.rs:259-304`), restated step by
    step: unsigned c-bit windows taken from the 32-byte LE repr (an 8-byte
    little-endian load at byte `skip_bits/8`, shifted, masked, :271-281),
    2^c-1 buckets, windows processed top-down with c doublings between them,
    zero digits skipped (:293), buckets folded by the running-sum trick
    (:298-302).  Returns the affine normalisation of the reference's
    projective result."""
    n = len(scalars)
    if n == 0:
        raise ValueError("empty MSM (reference indexes scalars[0]: msm.rs:265)")
    assert n == len(points)  # assert_eq! at msm.rs:309
    reprs = [fe_to_bytes(s) for s in scalars]
    num_bits = 256
    c = pippenger_window_size(n)
    num_buckets = (1 << c) - 1

    def windowed(idx, b):
        skip_bits = idx * c
        skip_bytes = skip_bits // 8
        v = int.from_bytes(b[skip_bytes:skip_bytes + 8], "little")
        return (v >> (skip_bits - skip_bytes * 8)) & num_buckets

    num_window = -(-num_bits // c)
    result = None
    for idx in reversed(range(num_window)):
        for _ in range(c):
            result = g1_double(result)
        buckets = [None] * num_buckets
        for b, pt in zip(reprs, points):
            d = windowed(idx, b)
            if d != 0:
                buckets[d - 1] = g1_add(buckets[d - 1], pt)
        running = None
        for bk in reversed(buckets):
            running = g1_add(bk, running)
            result = g1_add(result, running)
    return result


# --------------------------------------------------------------------------
# Extension tower: Fq2 = Fq[u]/(u^2+1); Fq6 = Fq2[v]/(v^3-xi), xi = 9+u;
# Fq12 = Fq6[w]/(w^2-v).   
# --------------------------------------------------------------------------
class Fq2:
    __slots__ = ("a", "b")

    def __init__(self, a, b=0):
        self.a = a % P
        self.b = b % P

    def __add__(self, o):
        return Fq2(self.a + o.a, self.b + o.b)

    def __sub__(self, o):
        return Fq2(self.a - o.a, self.b - o.b)

    def __neg__(self):
        return Fq2(-self.a, -self.b)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fq2(self.a * o, self.b * o)
        return Fq2(self.a * o.a - self.b * o.b, self.a * o.b + self.b * o.a)

    def __eq__(self, o):
        return self.a == o.a and self.b == o.b

    def is_zero(self):
        return self.a == 0 and self.b == 0

    def conj(self):
        return Fq2(self.a, -self.b)

    def inv(self):
        d = pow(self.a * self.a + self.b * self.b, -1, P)
        return Fq2(self.a * d, -self.b * d)

    def mul_xi(self):
        # (a + b u)(9 + u) = 9a - b + (a + 9b) u
        return Fq2(9 * self.a - self.b, self.a + 9 * self.b)

    def pow(self, e):
        res, base = Fq2(1), self
        while e:
            if e & 1:
                res = res * base
            base = base * base
            e >>= 1
        return res

    def to_bytes(self):
        return fe_to_bytes(self.a) + fe_to_bytes(self.b)

    @staticmethod
    def from_bytes(b):
        return Fq2(fe_from_bytes(b[:32]), fe_from_bytes(b[32:64]))

    def __repr__(self):
        return "Fq2(%#x, %#x)" % (self.a, self.b)


XI = Fq2(9, 1)
FQ2_ZERO, FQ2_ONE = Fq2(0), Fq2(1)


class Fq6:
    __slots__ = ("c0", "c1", "c2")

    def __init__(self, c0, c1, c2):
        self.c0, self.c1, self.c2 = c0, c1, c2

    def __add__(self, o):
        return Fq6(self.c0 + o.c0, self.c1 + o.c1, self.c2 + o.c2)

    def __sub__(self, o):
        return Fq6(self.c0 - o.c0, self.c1 - o.c1, self.c2 - o.c2)

    def __neg__(self):
        return Fq6(-self.c0, -self.c1, -self.c2)

    def __mul__(self, o):
        a0, a1, a2, b0, b1, b2 = self.c0, self.c1, self.c2, o.c0, o.c1, o.c2
        return Fq6(
            a0 * b0 + (a1 * b2 + a2 * b1).mul_xi(),
            a0 * b1 + a1 * b0 + (a2 * b2).mul_xi(),
            a0 * b2 + a1 * b1 + a2 * b0,
        )

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1 and self.c2 == o.c2

    def mul_v(self):
        return Fq6(self.c2.mul_xi(), self.c0, self.c1)

    def inv(self):
        a0, a1, a2 = self.c0, self.c1, self.c2
        t0 = a0 * a0 - (a1 * a2).mul_xi()
        t1 = (a2 * a2).mul_xi() - a0 * a1
        t2 = a1 * a1 - a0 * a2
        d = (a0 * t0 + (a2 * t1 + a1 * t2).mul_xi()).inv()
        return Fq6(t0 * d, t1 * d, t2 * d)


FQ6_ZERO = Fq6(FQ2_ZERO, FQ2_ZERO, FQ2_ZERO)
FQ6_ONE = Fq6(FQ2_ONE, FQ2_ZERO, FQ2_ZERO)


class Fq12:
    __slots__ = ("c0", "c1")

    def __init__(self, c0, c1):
        self.c0, self.c1 = c0, c1

    def __mul__(self, o):
        a0, a1, b0, b1 = self.c0, self.c1, o.c0, o.c1
        return Fq12(a0 * b0 + (a1 * b1).mul_v(), a0 * b1 + a1 * b0)

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1

    def conj(self):
        return Fq12(self.c0, -self.c1)

    def inv(self):
        d = (self.c0 * self.c0 - (self.c1 * self.c1).mul_v()).inv()
        return Fq12(self.c0 * d, -(self.c1 * d))

    def pow(self, e):
        res, base = FQ12_ONE, self
        while e:
            if e & 1:
                res = res * base
            base = base * base
            e >>= 1
        return res

    def is_one(self):
        return self == FQ12_ONE

    def coeffs(self):
        """12 Fq coefficients in tower order c0.c0.a, c0.c0.b, c0.c1.a, ... c1.c2.b."""
        out = []
        for c6 in (self.c0, self.c1):
            for c2 in (c6.c0, c6.c1, c6.c2):
                out += [c2.a, c2.b]
        return out

    def to_bytes(self):
        return b"".join(fe_to_bytes(c) for c in self.coeffs())


FQ12_ONE = Fq12(FQ6_ONE, FQ6_ZERO)


# --------------------------------------------------------------------------
# G2: D-type sextic twist  y^2 = x^3 + 3/xi  over Fq2.  None = identity.
# Generator: the EIP-197 / halo2curves G2 generator.
# --------------------------------------------------------------------------
B2 = Fq2(3) * XI.inv()
G2_GEN = (
    Fq2(
        10857046999023057135944570762232829481370756359578518086990519993285655852781,
        11559732032986387107991004021392285783925812861821192530917403151452391805634,
    ),
    Fq2(
        8495653923123431417604973247489272438418190587263600148770280649306958101930,
        4082367875863433681332203403145435568316851327593401208105741076214120093531,
    ),
)


def g2_is_on_curve(q):
    if q is None:
        return True
    x, y = q
    return y * y == x * x * x + B2


def g2_neg(q):
    return None if q is None else (q[0], -q[1])


def g2_double(q):
    if q is None:
        return None
    x, y = q
    if y.is_zero():
        return None
    lam = (x * x) * 3 * (y * 2).inv()
    x3 = lam * lam - x * 2
    return (x3, lam * (x - x3) - y)


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2).is_zero():
            return None
        return g2_double(a)
    lam = (y2 - y1) * (x2 - x1).inv()
    x3 = lam * lam - x1 - x2
    return (x3, lam * (x1 - x3) - y1)


def g2_mul(q, k):
    acc = None
    for bit in bin(int(k))[2:] if k else "":
        acc = g2_double(acc)
        if bit == "1":
            acc = g2_add(acc, q)
    return acc


def g2_to_bytes(q):
    if q is None:
        return b"\x00" * 128
    return q[0].to_bytes() + q[1].to_bytes()


def g2_from_bytes(b):
    x, y = Fq2.from_bytes(b[:64]), Fq2.from_bytes(b[64:128])
    if x.is_zero() and y.is_zero():
        return None
    return (x, y)


# Frobenius constants on the twist: pi(x', y') = (conj(x') * xi^((p-1)/3),
# conj(y') * xi^((p-1)/2)).
GAMMA_12 = XI.pow((P - 1) // 3)
GAMMA_13 = XI.pow((P - 1) // 2)
GAMMA_22 = XI.pow((P * P - 1) // 3)
GAMMA_23 = XI.pow((P * P - 1) // 2)


def g2_frobenius(q):
    return (q[0].conj() * GAMMA_12, q[1].conj() * GAMMA_13)


def g2_frobenius2(q):
    return (q[0] * GAMMA_22, q[1] * GAMMA_23)


# --------------------------------------------------------------------------
# Optimal ate pairing.  Untwist psi(x', y') = (x' w^2, y' w^3), w^6 = xi.
# A line through T with twist-slope lam evaluated at P = (xP, yP) in G1 is
#   l(P) = yP - (lam xP) w + (lam xT - yT) w^3
# and in the tower 1 <-> c0.c0, w <-> c1.c0, w^3 = v w <-> c1.c1.
# --------------------------------------------------------------------------
def _line(lam, t, p):
    xp, yp = p
    return Fq12(
        Fq6(Fq2(yp), FQ2_ZERO, FQ2_ZERO),
        Fq6(-(lam * xp), lam * t[0] - t[1], FQ2_ZERO),
    )


def _line_double(t, p):
    lam = (t[0] * t[0]) * 3 * (t[1] * 2).inv()
    return _line(lam, t, p), g2_double(t)


def _line_add(t, q, p):
    # callers never hit t == +-q for points of prime order r inside the loop
    lam = (q[1] - t[1]) * (q[0] - t[0]).inv()
    return _line(lam, t, p), g2_add(t, q)


def miller_loop(pairs):
    """Product of Miller functions f_{6x+2,Q}(P) * (two Frobenius lines) over
    (P in G1, Q in G2) pairs, squarings shared -- the definition of
    `MultiMillerLoop::multi_miller_loop` used at reference
    `snark-verifier/src/pcs/kzg/decider.rs:76`.  Pairs with an identity
    member contribute 1."""
    pairs = [(p, q) for p, q in pairs if p is not None and q is not None]
    f = FQ12_ONE
    ts = [q for _, q in pairs]
    bits = bin(ATE_LOOP)[3:]
    for bit in bits:
        f = f * f
        for i, (p, q) in enumerate(pairs):
            l, ts[i] = _line_double(ts[i], p)
            f = f * l
        if bit == "1":
            for i, (p, q) in enumerate(pairs):
                l, ts[i] = _line_add(ts[i], q, p)
                f = f * l
    for i, (p, q) in enumerate(pairs):
        q1 = g2_frobenius(q)
        q2 = g2_neg(g2_frobenius2(q))
        l, ts[i] = _line_add(ts[i], q1, p)
        f = f * l
        l, ts[i] = _line_add(ts[i], q2, p)
        f = f * l
    return f


FINAL_EXP = (P**12 - 1) // R


def final_exponentiation(f):
    """f^((p^12-1)/r), by plain square-and-multiply on the exact exponent."""
    return f.pow(FINAL_EXP)


def pairing(p, q):
    return final_exponentiation(miller_loop([(p, q)]))


def kzg_decide(lhs, rhs, g2, s_g2):
    """`KzgAs::decide` (reference `snark-verifier/src/pcs/kzg/decider.rs:70-82`):
    accept iff e(lhs, g2) * e(rhs, -s_g2) == 1 in Gt."""
    f = miller_loop([(lhs, g2), (rhs, g2_neg(s_g2))])
    return final_exponentiation(f).is_one()
