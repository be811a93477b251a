"""Inner-product-argument PCS + accumulation scheme oracle -- TEST INFRASTRUCTURE ONLY.

Python restatement (big-int scalars, `oracle/bn254.py` for the group) of the
reference's IPA layer, the second consumer of the MSM hot path
(`IpaAs::decide` is ONE `util::msm::multi_scalar_multiplication` of 2^k terms):

  h_eval / h_coeffs          <- snark-verifier/src/pcs/ipa.rs:391-421
  IpaProvingKey.commit       <- snark-verifier/src/pcs/ipa.rs:199-231
  ipa_create_proof           <- snark-verifier/src/pcs/ipa.rs:39-124      (prover: makes the fixtures)
  ipa_read_proof             <- snark-verifier/src/pcs/ipa.rs:320-356
  ipa_succinct_verify        <- snark-verifier/src/pcs/ipa.rs:139-180
  ipa_as_create_proof        <- snark-verifier/src/pcs/ipa/accumulation.rs:148-226
  ipa_as_read_proof / verify <- snark-verifier/src/pcs/ipa/accumulation.rs:41-146
  ipa_decide                 <- snark-verifier/src/pcs/ipa/decider.rs:47-55

The scheme is generic over the curve (`C: CurveAffine`); the reference's own
tests instantiate it on pallas (pcs/ipa.rs:434-466, accumulation.rs:240-290),
this restatement and the C++ mirror (`snark-verifier_amd/host/ipa.hpp`) on
BN254 G1, the curve the device kernels are built for.

PARITY UNPINNED: the reference's IPA tests draw everything from `OsRng` and
check accept only; there are no fixtures.  Pinned by construction instead: an
honest proof must pass the succinct check and `decide`, and
h_eval(xi, z) must equal the evaluation at z of the polynomial h_coeffs(xi, 1).
"""
import bn254 as O

R = O.R
_FAST = True  # the C restatement (oracle/c) is BN254 only


def use_curve(mod):
    """Run the curve-generic part (everything above the Bgh19 section) on another curve module with
    the interface of `oracle/bn254.py` -- `oracle/pallas.py`, the curve of the reference's own IPA
    tests.  `use_curve(bn254)` switches back."""
    global O, R, _FAST
    O, R = mod, mod.R
    _FAST = mod.__name__ == "bn254"



def _inv(a):
    return pow(a % R, -1, R)


def inner_product(a, b):
    return sum(x * y for x, y in zip(a, b)) % R


def poly_eval(coeffs, z):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * z + c) % R
    return acc


# ipa.rs:391-403: prod_i (1 + xi_{k-1-i} z^(2^i))
def h_eval(xi, z):
    out, zp = 1, z % R
    for x in reversed(xi):
        out = out * ((zp * x + 1) % R) % R
        zp = zp * zp % R
    return out


# ipa.rs:405-421
def h_coeffs(xi, scalar=1):
    assert len(xi) > 0
    coeffs = [0] * (1 << len(xi))
    coeffs[0] = scalar % R
    for i, x in enumerate(reversed(xi)):
        ln = 1 << i
        for j in range(ln):
            coeffs[ln + j] = coeffs[j] * x % R
    return coeffs


def _msm(scalars, points):
    """`util::msm::multi_scalar_multiplication`; the C restatement when it is built (much faster)."""
    assert len(scalars) == len(points) and len(scalars) > 0
    if not _FAST:
        return O.g1_msm_pippenger([s % R for s in scalars], points)
    try:
        import coracle as C

        sb = b"".join(O.fe_to_bytes(s % R) for s in scalars)
        pb = b"".join(O.g1_to_bytes(p) for p in points)
        return O.g1_from_bytes(C.msm_pippenger(sb, pb, 1))
    except (ImportError, OSError):
        return O.g1_msm_naive([s % R for s in scalars], points)


def _mul(pt, k):
    if not _FAST:
        return O.g1_mul(pt, k % R)
    try:
        import coracle as C

        return O.g1_from_bytes(C.g1_mul(O.g1_to_bytes(pt), O.fe_to_bytes(k % R)))
    except (ImportError, OSError):
        return O.g1_mul(pt, k % R)


class IpaProvingKey:
    """ipa.rs:183-231 (`domain` enters only through k and n = 2^k)."""

    def __init__(self, k, g, h, s=None):
        assert len(g) == 1 << k
        self.k, self.g, self.h, self.s = k, list(g), h, s

    def zk(self):
        return self.s is not None

    def commit(self, poly, omega=None):
        c = _msm(poly, self.g)
        assert (self.s is None) == (omega is None)
        if self.s is not None:
            c = O.g1_add(c, _mul(self.s, omega))
        return c


# ipa.rs:39-124.  `rng()` returns a fresh scalar (plays `C::Scalar::random`).
def ipa_create_proof(pk, p, z, omega, transcript, rng):
    n = 1 << pk.k
    p_prime = [c % R for c in p]
    assert len(p_prime) == n
    if pk.zk():
        p_bar = [rng() for _ in range(n)]
        p_bar[0] = (p_bar[0] - poly_eval(p_bar, z)) % R
        omega_bar = rng()
        c_bar = pk.commit(p_bar, omega_bar)
        transcript.write_ec_point(c_bar)
        alpha = transcript.squeeze_challenge()
        omega_prime = (omega + alpha * omega_bar) % R
        transcript.write_scalar(omega_prime)
        p_prime = [(a + alpha * b) % R for a, b in zip(p_prime, p_bar)]
    xi_0 = transcript.squeeze_challenge()
    h_prime = _mul(pk.h, xi_0)
    bases, coeffs = list(pk.g), p_prime
    zs = [pow(z, i, R) for i in range(n)]
    xi = []
    for i in range(pk.k):
        half = 1 << (pk.k - i - 1)
        l_i = O.g1_add(_msm(coeffs[half:], bases[:half]), _mul(h_prime, inner_product(coeffs[half:], zs[:half])))
        r_i = O.g1_add(_msm(coeffs[:half], bases[half:]), _mul(h_prime, inner_product(coeffs[:half], zs[half:])))
        transcript.write_ec_point(l_i)
        transcript.write_ec_point(r_i)
        xi_i = transcript.squeeze_challenge()
        xi_i_inv = _inv(xi_i)
        bases = [O.g1_add(bases[j], _mul(bases[half + j], xi_i)) for j in range(half)]
        coeffs = [(coeffs[j] + xi_i_inv * coeffs[half + j]) % R for j in range(half)]
        zs = [(zs[j] + xi_i * zs[half + j]) % R for j in range(half)]
        xi.append(xi_i)
    transcript.write_ec_point(bases[0])
    transcript.write_scalar(coeffs[0])
    return xi, bases[0]  # IpaAccumulator::new(xi, bases[0])


# ipa.rs:320-356
def ipa_read_proof(zk, k, transcript):
    c_bar_alpha = omega_prime = None
    if zk:
        c_bar = transcript.read_ec_point()
        c_bar_alpha = (c_bar, transcript.squeeze_challenge())
        omega_prime = transcript.read_scalar()
    xi_0 = transcript.squeeze_challenge()
    rounds = []
    for _ in range(k):
        l = transcript.read_ec_point()
        r = transcript.read_ec_point()
        rounds.append((l, r, transcript.squeeze_challenge()))
    u = transcript.read_ec_point()
    c = transcript.read_scalar()
    return dict(c_bar_alpha=c_bar_alpha, omega_prime=omega_prime, xi_0=xi_0, rounds=rounds, u=u, c=c)


class IpaError(Exception):
    """`Error::AssertionFailure`"""


# ipa.rs:139-180.  `commitment` = [(scalar, point), ...] (an `Msm` without constant).
def ipa_succinct_verify(h, s, commitment, z, ev, proof):
    xi = [r[2] for r in proof["rounds"]]
    xi_inv = [_inv(x) for x in xi]
    terms = list(commitment)
    if s is not None:
        c_bar, alpha = proof["c_bar_alpha"]
        terms += [(alpha, c_bar), ((-proof["omega_prime"]) % R, s)]
    else:
        assert proof["c_bar_alpha"] is None and proof["omega_prime"] is None
    terms.append((proof["xi_0"] * ev % R, h))  # h_prime * eval
    for (l, r, x), xinv in zip(proof["rounds"], xi_inv):
        terms += [(xinv, l), (x, r)]
    lhs = _msm([t[0] for t in terms], [t[1] for t in terms])
    v_prime = h_eval(xi, z) * proof["c"] % R
    rhs = _msm([proof["c"], proof["xi_0"] * v_prime % R], [proof["u"], h])
    if lhs != rhs:
        raise IpaError("C_k == c[U] + v'[H']")
    return xi, proof["u"]


# decider.rs:47-55
def ipa_decide(g, acc):
    xi, u = acc
    return u == _msm(h_coeffs(xi, 1), g)


# accumulation.rs:148-226
def ipa_as_create_proof(pk, instances, transcript, rng):
    assert len(instances) > 1
    n = 1 << pk.k
    a_b_u = omega = None
    if pk.zk():
        a, b = rng(), rng()
        u = O.g1_add(_mul(pk.g[1], a), _mul(pk.g[0], b))
        transcript.write_scalar(a)
        transcript.write_scalar(b)
        transcript.write_ec_point(u)
        a_b_u = (a, b, u)
        omega = rng()
        transcript.write_scalar(omega)
    for xi, u in instances:
        for x in xi:
            transcript.common_scalar(x)
        transcript.common_ec_point(u)
    alpha = transcript.squeeze_challenge()
    z = transcript.squeeze_challenge()
    hs = [h_coeffs(xi, 1) for xi, _ in instances]
    if a_b_u is not None:
        hs.append([a_b_u[1], a_b_u[0]] + [0] * (n - 2))
    h = [0] * n
    pw = 1
    for hc in hs:
        h = [(x + pw * y) % R for x, y in zip(h, hc)]
        pw = pw * alpha % R
    return ipa_create_proof(pk, h, z, omega, transcript, rng)


# accumulation.rs:98-146
def ipa_as_read_proof(zk, k, instances, transcript):
    assert len(instances) > 1
    a_b_u = omega = None
    if zk:
        a = transcript.read_scalar()
        b = transcript.read_scalar()
        a_b_u = (a, b, transcript.read_ec_point())
        omega = transcript.read_scalar()
    for xi, u in instances:
        for x in xi:
            transcript.common_scalar(x)
        transcript.common_ec_point(u)
    alpha = transcript.squeeze_challenge()
    z = transcript.squeeze_challenge()
    return dict(a_b_u=a_b_u, omega=omega, alpha=alpha, z=z, ipa=ipa_read_proof(zk, k, transcript))


# accumulation.rs:41-79
def ipa_as_verify(h, s, instances, proof):
    z = proof["z"]
    us = [u for _, u in instances]
    hv = [h_eval(xi, z) for xi, _ in instances]
    if proof["a_b_u"] is not None:
        a, b, u = proof["a_b_u"]
        us.append(u)
        hv.append((a * z + b) % R)
    pw, powers = 1, []
    for _ in us:
        powers.append(pw)
        pw = pw * proof["alpha"] % R
    c = list(zip(powers, us))
    if proof["omega"] is not None:
        c.append((proof["omega"], s))
    v = sum(p * x for p, x in zip(powers, hv)) % R
    return ipa_succinct_verify(h, s, c, z, v, proof["ipa"])


# ---------------------------------------------------------------------------
# Bgh19: the multi-open scheme of halo2's IPA backend
#   bgh19_read_proof     <- snark-verifier/src/pcs/ipa/multiopen/bgh19.rs:113-153
#   query sets / coeffs  <- bgh19.rs:155-250, 313-399  (same grouping as bdfg21.rs:121-171)
#   bgh19_verify         <- bgh19.rs:48-96
#   bgh19_create_proof   <- no reference code (the prover lives in halo2_proofs): written from the
#                           verifier's equations, transcript order as `Bgh19Proof::read`
# ---------------------------------------------------------------------------
import kzg as K  # noqa: E402  (Msm, the shared query-set grouping)


def bgh19_read_proof(k, queries, transcript):
    x_1 = transcript.squeeze_challenge()
    x_2 = transcript.squeeze_challenge()
    f = transcript.read_ec_point()
    x_3 = transcript.squeeze_challenge()
    q_evals = [transcript.read_scalar() for _ in K.bdfg21_query_sets(queries)]
    x_4 = transcript.squeeze_challenge()
    s = transcript.read_ec_point()
    xi = transcript.squeeze_challenge()
    z = transcript.squeeze_challenge()
    rounds = []
    for _ in range(k):
        l = transcript.read_ec_point()
        r = transcript.read_ec_point()
        rounds.append((l, r, transcript.squeeze_challenge()))
    c = transcript.read_scalar()
    blind = transcript.read_scalar()
    g = transcript.read_ec_point()
    ipa = dict(c_bar_alpha=(s, xi), omega_prime=blind, xi_0=z, rounds=rounds, u=g, c=c)  # bgh19.rs:150
    return dict(x_1=x_1, x_2=x_2, f=f, x_3=x_3, q_evals=q_evals, x_4=x_4, ipa=ipa)


def bgh19_query_set_coeffs(sets, x, x_3):
    """bgh19.rs:217-250 + QuerySetCoeff (:313-399), after its two batch inversions."""
    size = max([len(st["shifts"]) for st in sets] + [2])
    px = K.powers(x, size)
    out = []
    for st in sets:
        shifts = st["shifts"]
        ell = []
        for j, sj in enumerate(shifts):
            acc = 1
            for i, si in enumerate(shifts):
                if i != j:
                    acc = acc * (sj - si) % R
            ell.append(acc)
        xk1 = px[len(shifts) - 1]
        bary = [_inv((e * xk1 * x_3 - e * s * xk1 * px[1]) % R) for s, e in zip(shifts, ell)]
        den = 1
        for s in shifts:
            den = den * (x_3 - x * s) % R
        out.append(dict(eval_coeffs=bary, r_eval_coeff=_inv(sum(bary) % R), f_eval_coeff=_inv(den)))
    return out


def bgh19_final_msm(g0, commitments, x, queries, proof):
    """The `p` of bgh19.rs:61-93: the commitment whose opening at x_3 must be 0."""
    sets = K.bdfg21_query_sets(queries)
    coeffs = bgh19_query_set_coeffs(sets, x, proof["x_3"])
    px1 = K.powers(proof["x_1"], max(len(st["polys"]) for st in sets))
    f_evals = []
    for st, co, q_eval in zip(sets, coeffs, proof["q_evals"]):
        r_evals = [sum(c * e for c, e in zip(co["eval_coeffs"], evals)) % R * co["r_eval_coeff"] % R for evals in st["evals"]]
        r_eval = sum(a * b for a, b in zip(reversed(r_evals), px1)) % R
        f_evals.append((q_eval - r_eval) * co["f_eval_coeff"] % R)
    px2 = K.powers(proof["x_2"], len(sets))
    f_eval = sum(a * b for a, b in zip(px2, reversed(f_evals))) % R
    terms = [K.Msm.base(proof["f"]) - K.Msm.const(f_eval)]
    for st, q_eval in zip(sets, proof["q_evals"]):
        m = K.Msm.sum([commitments[poly] * pw for poly, pw in zip(reversed(st["polys"]), px1)]) - K.Msm.const(q_eval)
        terms.append(m)
    px4 = K.powers(proof["x_4"], len(sets) + 1)
    total = K.Msm.sum([m * pw for m, pw in zip(terms, reversed(px4))])
    const, total.constant = total.constant, None  # `.split()`
    if const is not None:
        total = total + K.Msm.base(g0) * const
    return total


def bgh19_verify(g0, h, s, commitments, x, queries, proof):
    p = bgh19_final_msm(g0, commitments, x, queries, proof)
    return ipa_succinct_verify(h, s, list(zip(p.scalars, p.bases)), proof["x_3"], 0, proof["ipa"])


# ---- prover (test fixtures only) -------------------------------------------------------------
def _poly_add_scaled(a, b, f):
    n = max(len(a), len(b))
    a, b = a + [0] * (n - len(a)), b + [0] * (n - len(b))
    return [(x + f * y) % R for x, y in zip(a, b)]


def _poly_div_linear(p, a):
    """p(X) / (X - a), exact."""
    out, carry = [0] * (len(p) - 1), 0
    for i in range(len(p) - 1, 0, -1):
        carry = (p[i] + carry * a) % R
        out[i - 1] = carry
    assert (p[0] + carry * a) % R == 0, "not divisible"
    return out


def _interpolate(pts, vals):
    res = [0]
    for j, (pj, vj) in enumerate(zip(pts, vals)):
        num, den = [1], 1
        for i, pi in enumerate(pts):
            if i != j:
                num = [(a - pi * b) % R for a, b in zip([0] + num, num + [0])]
                den = den * (pj - pi) % R
        res = _poly_add_scaled(res, num, vj * _inv(den) % R)
    return res


def bgh19_create_proof(pk, polys, blinds, x, queries, transcript, rng):
    """`queries` = [(poly, shift, eval)] with eval = polys[poly](x * shift); commitments are
    pk.commit(polys[j], blinds[j]).  Writes what `Bgh19Proof::read` reads, in that order."""
    assert pk.zk()
    n = 1 << pk.k
    sets = K.bdfg21_query_sets(queries)
    x_1 = transcript.squeeze_challenge()
    x_2 = transcript.squeeze_challenge()
    px1 = K.powers(x_1, max(len(st["polys"]) for st in sets))
    qs, q_blinds, fs = [], [], []
    for st in sets:
        q, qb = [0] * n, 0
        for poly, pw in zip(reversed(st["polys"]), px1):  # QuerySet::msm order (bgh19.rs:268-273)
            q = _poly_add_scaled(q, polys[poly], pw)
            qb = (qb + pw * blinds[poly]) % R
        pts = [x * sh % R for sh in st["shifts"]]
        r = _interpolate(pts, [poly_eval(q, p) for p in pts])
        fi = _poly_add_scaled(q, r, R - 1)
        for p in pts:
            fi = _poly_div_linear(fi, p)
        qs.append(q)
        q_blinds.append(qb)
        fs.append(fi)
    f = [0] * n
    for fi, pw in zip(reversed(fs), K.powers(x_2, len(sets))):  # f_eval pairs x_2^j with f_evals.rev()
        f = _poly_add_scaled(f, fi, pw)[:n]
    f_blind = rng()
    transcript.write_ec_point(pk.commit(f, f_blind))
    x_3 = transcript.squeeze_challenge()
    for q in qs:
        transcript.write_scalar(poly_eval(q, x_3))
    x_4 = transcript.squeeze_challenge()
    px4 = list(reversed(K.powers(x_4, len(sets) + 1)))
    p = [c * px4[0] % R for c in f]
    omega = f_blind * px4[0] % R
    p[0] = (p[0] - poly_eval(f, x_3) * px4[0]) % R
    for q, qb, pw in zip(qs, q_blinds, px4[1:]):
        p = _poly_add_scaled(p, q, pw)
        p[0] = (p[0] - poly_eval(q, x_3) * pw) % R
        omega = (omega + qb * pw) % R
    assert poly_eval(p, x_3) == 0
    # the IPA opening of p at x_3 (value 0), zero-knowledge form, in Bgh19's transcript order
    p_bar = [rng() for _ in range(n)]
    p_bar[0] = (p_bar[0] - poly_eval(p_bar, x_3)) % R
    omega_bar = rng()
    transcript.write_ec_point(pk.commit(p_bar, omega_bar))  # `s`
    alpha = transcript.squeeze_challenge()                   # `xi`
    xi_0 = transcript.squeeze_challenge()                    # `z`
    coeffs = _poly_add_scaled(p, p_bar, alpha)
    omega_prime = (omega + alpha * omega_bar) % R
    h_prime = _mul(pk.h, xi_0)
    bases = list(pk.g)
    zs = [pow(x_3, i, R) for i in range(n)]
    for i in range(pk.k):
        half = 1 << (pk.k - i - 1)
        l_i = O.g1_add(_msm(coeffs[half:], bases[:half]), _mul(h_prime, inner_product(coeffs[half:], zs[:half])))
        r_i = O.g1_add(_msm(coeffs[:half], bases[half:]), _mul(h_prime, inner_product(coeffs[:half], zs[half:])))
        transcript.write_ec_point(l_i)
        transcript.write_ec_point(r_i)
        xi_i = transcript.squeeze_challenge()
        xi_i_inv = _inv(xi_i)
        bases = [O.g1_add(bases[j], _mul(bases[half + j], xi_i)) for j in range(half)]
        coeffs = [(coeffs[j] + xi_i_inv * coeffs[half + j]) % R for j in range(half)]
        zs = [(zs[j] + xi_i * zs[half + j]) % R for j in range(half)]
    transcript.write_scalar(coeffs[0])
    transcript.write_scalar(omega_prime)
    transcript.write_ec_point(bases[0])
