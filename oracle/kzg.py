"""KZG accumulation layer oracle -- TEST INFRASTRUCTURE ONLY.

Python restatement (big-int Fr, `oracle/bn254.py` for G1) of the reference's
host-side algebra on the hot path, step by step, so the C++ mirror in
`snark-verifier_amd/host/` can be checked against it:

  Msm                <- snark-verifier/src/util/msm.rs:20-226
  kzg_as_verify      <- snark-verifier/src/pcs/kzg/accumulation.rs:41-63
  gwc19_verify       <- snark-verifier/src/pcs/kzg/multiopen/gwc19.rs:45-82,124-160
  bdfg21_verify      <- snark-verifier/src/pcs/kzg/multiopen/bdfg21.rs:51-83,121-371
  limbs_from_repr /  <- snark-verifier/src/pcs/kzg/accumulator.rs:57-81,
  fe_to_limbs           snark-verifier/src/util/arithmetic.rs:270-298

PARITY UNPINNED (no reference fixtures exist for this path and the reference
cannot be run here: SURVEY.md 8c); pinned by construction instead: the
synthetic instances below are VALID openings under a toy SRS secret, so the
accumulators every scheme returns must satisfy e(lhs, g2) = e(rhs, s g2).
"""
import bn254 as O

R = O.R


def use_curve(mod):
    """Run the curve-generic parts (Msm, powers, the query-set grouping) over another curve module with
    the interface of oracle/bn254.py (oracle/pallas.py); the KZG schemes themselves stay BN254."""
    global O, R
    O, R = mod, mod.R


def fr_inv(a):
    return pow(a % R, -1, R)


def powers(x, n):
    """`LoadedScalar::powers` (reference loader.rs:71-78): 1, x, x^2, ... (n items)."""
    out = [1]
    for _ in range(n - 1):
        out.append(out[-1] * x % R)
    return out[:n]


class Msm:
    """Deferred `constant*G + sum scalar_i * base_i` (msm.rs:20-24).  Bases are
    affine points (tuples / None); `push` merges EQUAL bases (msm.rs:109-116)."""

    def __init__(self, constant=None, scalars=None, bases=None):
        self.constant = constant
        self.scalars = list(scalars or [])
        self.bases = list(bases or [])

    @staticmethod
    def const(c):
        return Msm(constant=c % R)

    @staticmethod
    def base(pt):
        return Msm(scalars=[1], bases=[pt])

    def copy(self):
        return Msm(self.constant, self.scalars, self.bases)

    def scale(self, f):
        if self.constant is not None:
            self.constant = self.constant * f % R
        self.scalars = [s * f % R for s in self.scalars]

    def push(self, scalar, base):
        for i, b in enumerate(self.bases):
            if b == base:
                self.scalars[i] = (self.scalars[i] + scalar) % R
                return
        self.scalars.append(scalar % R)
        self.bases.append(base)

    def extend(self, other):
        if other.constant is not None:
            self.constant = other.constant if self.constant is None else (self.constant + other.constant) % R
        for s, b in zip(other.scalars, other.bases):
            self.push(s, b)

    def __add__(self, o):
        r = self.copy()
        r.extend(o)
        return r

    def __neg__(self):
        return Msm(None if self.constant is None else (-self.constant) % R, [(-s) % R for s in self.scalars], self.bases)

    def __sub__(self, o):
        return self + (-o)

    def __mul__(self, f):
        r = self.copy()
        r.scale(f % R)
        return r

    @staticmethod
    def sum(msms):
        msms = list(msms)
        if not msms:
            return Msm()
        acc = msms[0].copy()
        for m in msms[1:]:
            acc.extend(m)
        return acc

    def pairs(self, gen):
        """The (scalar, base) list `evaluate` hands to the loader (msm.rs:81-98):
        constant*gen first, then the terms in order."""
        out = []
        if self.constant is not None:
            assert gen is not None, "constant without generator (reference panics: msm.rs:85,93)"
            out.append((self.constant, gen))
        out += list(zip(self.scalars, self.bases))
        return out

    def evaluate(self, gen):
        prs = self.pairs(gen)
        return O.g1_msm_naive([s for s, _ in prs], [b for _, b in prs])


# --------------------------------------------------------------------------
def kzg_as_verify(accumulators, r, blind=None):
    """`KzgAs::verify` (accumulation.rs:41-63): random linear combination of the
    accumulators with powers of r; the optional zk blind pair goes last."""
    lhs = [a[0] for a in accumulators]
    rhs = [a[1] for a in accumulators]
    if blind is not None:
        lhs.append(blind[0])
        rhs.append(blind[1])
    pw = powers(r, len(lhs))
    out = []
    for bases in (lhs, rhs):
        out.append(Msm.sum(Msm.base(b) * p for b, p in zip(bases, pw)).evaluate(None))
    return tuple(out)


# --------------------------------------------------------------------------
def gwc19_query_sets(queries):
    """group by shift, first-seen order (gwc19.rs:142-160). queries: (poly, shift, eval)."""
    sets = []
    for poly, shift, ev in queries:
        for st in sets:
            if st["shift"] == shift:
                st["polys"].append(poly)
                st["evals"].append(ev)
                break
        else:
            sets.append({"shift": shift, "polys": [poly], "evals": [ev]})
    return sets


def gwc19_msms(g, commitments, z, queries, v, ws, u):
    """The two Msm the verifier evaluates (gwc19.rs:45-82)."""
    sets = gwc19_query_sets(queries)
    pu = powers(u, len(sets))
    pv = powers(v, max(len(st["polys"]) for st in sets))
    f = Msm.sum(
        Msm.sum((commitments[p] - Msm.const(e)) * pvi for p, e, pvi in zip(st["polys"], st["evals"], pv)) * pui
        for st, pui in zip(sets, pu)
    )
    z_omegas = [st["shift"] * z % R for st in sets]
    rhs = [Msm.base(w) * pui for w, pui in zip(ws, pu)]
    lhs = f + Msm.sum(uw * zo for uw, zo in zip(rhs, z_omegas))
    return lhs, Msm.sum(rhs)


def gwc19_verify(g, commitments, z, queries, v, ws, u):
    lhs, rhs = gwc19_msms(g, commitments, z, queries, v, ws, u)
    return lhs.evaluate(g), rhs.evaluate(g)


# --------------------------------------------------------------------------
def bdfg21_query_sets(queries):
    """bdfg21.rs:121-171: per poly the distinct shifts (first-seen), then group
    polys whose shift SETS are equal; evals re-ordered to the set's shift order."""
    poly_shifts = []
    for poly, shift, ev in queries:
        for ent in poly_shifts:
            if ent[0] == poly:
                if shift not in ent[1]:
                    ent[1].append(shift)
                    ent[2].append(ev)
                break
        else:
            poly_shifts.append((poly, [shift], [ev]))
    sets = []
    for poly, shifts, evals in poly_shifts:
        for st in sets:
            if set(st["shifts"]) == set(shifts):
                if poly not in st["polys"]:
                    st["polys"].append(poly)
                    st["evals"].append([evals[shifts.index(s)] for s in st["shifts"]])
                break
        else:
            sets.append({"shifts": shifts, "polys": [poly], "evals": [evals]})
    return sets


def bdfg21_msms(g, commitments, z, queries, mu, gamma, w, z_prime, w_prime):
    """bdfg21.rs:51-83 with query_set_coeffs (:173-223) and QuerySetCoeff (:267-371)."""
    sets = bdfg21_query_sets(queries)
    size = max([len(st["shifts"]) for st in sets] + [2])
    pz = powers(z, size)
    coeffs = []
    z_s_1 = None
    for st in sets:
        shifts = st["shifts"]
        ell = []
        for j, sj in enumerate(shifts):
            acc = 1
            for i, si in enumerate(shifts):
                if i != j:
                    acc = acc * (sj - si) % R
            ell.append(acc)
        zk1 = pz[len(shifts) - 1]
        bary = [fr_inv((e * zk1 * z_prime - e * s * zk1 * pz[1]) % R) for s, e in zip(shifts, ell)]
        z_s = 1
        for s in shifts:
            z_s = z_s * (z_prime - z * s) % R
        if z_s_1 is None:
            commitment_coeff = None
            r_eval_coeff = fr_inv(sum(bary) % R)
            z_s_1 = z_s
        else:
            commitment_coeff = z_s_1 * fr_inv(z_s) % R
            r_eval_coeff = commitment_coeff * fr_inv(sum(bary) % R) % R
        coeffs.append({"z_s": z_s, "eval_coeffs": bary, "commitment_coeff": commitment_coeff, "r_eval_coeff": r_eval_coeff})
    pmu = powers(mu, max(len(st["polys"]) for st in sets))
    msms = []
    for st, co in zip(sets, coeffs):
        terms = []
        for poly, evals, pm in zip(st["polys"], st["evals"], pmu):
            cm = commitments[poly] * co["commitment_coeff"] if co["commitment_coeff"] is not None else commitments[poly].copy()
            r_eval = sum(c * e for c, e in zip(co["eval_coeffs"], evals)) % R * co["r_eval_coeff"] % R
            terms.append((cm - Msm.const(r_eval)) * pm)
        msms.append(Msm.sum(terms))
    f = Msm.sum(m * pg for m, pg in zip(msms, powers(gamma, len(sets)))) - Msm.base(w) * coeffs[0]["z_s"]
    rhs = Msm.base(w_prime)
    lhs = f + rhs * z_prime
    return lhs, rhs


def bdfg21_verify(g, commitments, z, queries, mu, gamma, w, z_prime, w_prime):
    lhs, rhs = bdfg21_msms(g, commitments, z, queries, mu, gamma, w, z_prime, w_prime)
    return lhs.evaluate(g), rhs.evaluate(g)


# --------------------------------------------------------------------------
def fe_to_limbs(fe, limbs, bits):
    """`fe_to_limbs` (arithmetic.rs:286-298): little-endian limb order."""
    mask = (1 << bits) - 1
    return [(fe >> (bits * i)) & mask for i in range(limbs)]


def fe_from_limbs(ls, bits):
    """`fe_from_limbs` (arithmetic.rs:270-283); `fe_from_big` asserts the value is a
    canonical element (`F::from_repr(..).unwrap()`)."""
    v = sum(l << (bits * i) for i, l in enumerate(ls))
    if v >= O.P:
        raise ValueError("limbs encode a value >= p (reference panics: from_repr().unwrap())")
    return v


def limbs_from_repr(limb_values, limbs=4, bits=68):
    """`LimbsEncoding::from_repr` (accumulator.rs:57-81): 4*LIMBS Fr limbs ->
    KzgAccumulator; panics (here ValueError) when a point is off-curve."""
    assert len(limb_values) == 4 * limbs
    xs = [fe_from_limbs(limb_values[i * limbs:(i + 1) * limbs], bits) for i in range(4)]
    lhs, rhs = (xs[0], xs[1]), (xs[2], xs[3])
    for pt in (lhs, rhs):
        if not O.g1_is_on_curve(pt):
            raise ValueError("off-curve point (reference panics: from_xy().unwrap())")
    return lhs, rhs


def accumulator_to_limbs(acc, limbs=4, bits=68):
    out = []
    for pt in acc:
        x, y = (0, 0) if pt is None else pt
        out += fe_to_limbs(x, limbs, bits) + fe_to_limbs(y, limbs, bits)
    return out


# --------------------------------------------------------------------------
# Shape-faithful synthetic proofs (SURVEY.md 8c "Real proofs?"): commitments with
# known discrete logs under a toy SRS secret s; the opening points are SOLVED so
# that every accumulator is valid.  StandardPlonk shape: 17 commitments, 3
# rotations.
# --------------------------------------------------------------------------
def _mulg(k):
    import coracle as C

    return O.g1_from_bytes(C.g1_mul(O.g1_to_bytes(O.G1_GEN), O.fe_to_bytes(k % R)))


def standard_plonk_queries(rng, n_commit=17, shifts=None):
    """17 polys; every poly queried at rotation 0, a few also at 1 and `last`
    (permutation / lookup style) -> 3 query sets under GWC."""
    shifts = shifts or [1, rng.randrange(2, R), rng.randrange(2, R)]
    qs = []
    for p in range(n_commit):
        qs.append((p, shifts[0]))
    for p in (3, 4, 11):
        qs.append((p, shifts[1]))
    for p in (4, 12):
        qs.append((p, shifts[2]))
    return qs


def synth_gwc19_instance(rng, s, n_commit=17):
    g = O.G1_GEN
    c = [rng.randrange(R) for _ in range(n_commit)]
    cpts = [_mulg(ci) for ci in c]
    commitments = [Msm.base(p) for p in cpts]
    z, v, u = (rng.randrange(1, R) for _ in range(3))
    queries = [(p, sh, rng.randrange(R)) for p, sh in standard_plonk_queries(rng, n_commit)]
    sets = gwc19_query_sets(queries)
    pv = powers(v, max(len(st["polys"]) for st in sets))
    ws = []
    for st in sets:
        f = sum(pvi * (c[p] - e) for p, e, pvi in zip(st["polys"], st["evals"], pv)) % R
        ws.append(_mulg(f * fr_inv(s - z * st["shift"])))
    return {"g": g, "commitment_points": cpts, "commitments": commitments, "z": z, "queries": queries, "v": v, "ws": ws, "u": u}


def synth_bdfg21_instance(rng, s, n_commit=17):
    g = O.G1_GEN
    c = [rng.randrange(R) for _ in range(n_commit)]
    cpts = [_mulg(ci) for ci in c]
    commitments = [Msm.base(p) for p in cpts]
    z, mu, gamma, z_prime = (rng.randrange(1, R) for _ in range(4))
    queries = [(p, sh, rng.randrange(R)) for p, sh in standard_plonk_queries(rng, n_commit)]
    wlog = rng.randrange(R)
    w = _mulg(wlog)
    # dlog of f with W' = O, then W' = f / (s - z')
    lhs0, _ = bdfg21_msms(g, commitments, z, queries, mu, gamma, w, z_prime, None)
    dl = {pt: ci for pt, ci in zip(cpts, c)}
    dl[w] = wlog
    f = (lhs0.constant or 0)
    for sc, b in zip(lhs0.scalars, lhs0.bases):
        if b is None:
            continue  # the W' placeholder (scalar z') -- excluded from f
        f += sc * dl[b]
    w_prime = _mulg(f * fr_inv(s - z_prime))
    return {"g": g, "commitment_points": cpts, "commitments": commitments, "z": z, "queries": queries, "mu": mu,
            "gamma": gamma, "w": w, "z_prime": z_prime, "w_prime": w_prime}
