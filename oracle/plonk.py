"""PLONK succinct verifier front half -- oracle, TEST INFRASTRUCTURE ONLY.

Python restatement (big-int Fr) of SURVEY.md 8f row N1:

  Domain / Rotation          <- snark-verifier/src/util/arithmetic.rs:98-160
  Expression, Query,         <- snark-verifier/src/verifier/plonk/protocol.rs:274-420
  PlonkProtocol.langranges   <- protocol.rs:80-111
  CommonPolynomialEvaluation <- protocol.rs:196-272
  PlonkProof.read            <- snark-verifier/src/verifier/plonk/proof.rs:52-168
  .evaluations/.commitments/ <- proof.rs:170-349
  .queries
  PlonkSuccinctVerifier      <- snark-verifier/src/verifier/plonk.rs:58-92
  Gwc19Proof.read            <- pcs/kzg/multiopen/gwc19.rs:102-110
  Bdfg21Proof.read           <- pcs/kzg/multiopen/bdfg21.rs:95-117

plus `forge_proof`: a proof FORGER for a toy SRS whose secret is known.  Real
halo2 proofs cannot be produced here (no prover); with the secret s every
commitment is [c]G for a known c, so an opening witness W = [(c - e)/(s - z)]G
exists for ANY claimed evaluation e.  The forger walks the transcript in the
prover's order and emits bytes the verifier accepts under that SRS -- for any
protocol shape.  That gives end-to-end accept/reject tests of the whole
pipeline (proof bytes -> transcript -> expression evaluation -> MSMs ->
pairing) with the data shapes of the reference's StandardPlonk example.

PARITY UNPINNED against the Rust crate (cannot be built here).
"""
import bn254 as O
import kzg as K

R = O.R
Msm = K.Msm


def use_curve(mod):
    """Everything here is generic over the curve (`C: CurveAffine` in the reference); switch the whole
    oracle stack -- this module, kzg.py's Msm / query sets, ipa.py -- to `mod` (oracle/pallas.py), or back."""
    global O, R
    import ipa as I

    O, R = mod, mod.R
    K.use_curve(mod)
    I.use_curve(mod)


# ---------------------------------------------------------------- domain
def root_of_unity(k):
    """`root_of_unity` (arithmetic.rs:85-94): F::ROOT_OF_UNITY = g^((r-1)/2^S), squared S-k times
    (bn256 Fr: g = 7, S = 28; pasta: g = 5, S = 32)."""
    s_, g_ = O.TWO_ADICITY, O.MULT_GEN
    assert k <= s_
    w = pow(g_, (R - 1) >> s_, R)
    for _ in range(s_ - k):
        w = w * w % R
    return w


class Domain:
    def __init__(self, k, gen=None):
        self.k = k
        self.n = 1 << k
        self.gen = root_of_unity(k) if gen is None else gen
        self.n_inv = pow(self.n, -1, R)
        self.gen_inv = pow(self.gen, -1, R)

    def rotate_scalar(self, scalar, rot):  # arithmetic.rs:153-159
        if rot == 0:
            return scalar % R
        if rot > 0:
            return scalar * pow(self.gen, rot, R) % R
        return scalar * pow(self.gen_inv, -rot, R) % R


# ---------------------------------------------------------------- expressions
# ("const", f) ("identity",) ("lagrange", i) ("poly", poly, rot) ("challenge", i)
# ("neg", a) ("sum", a, b) ("prod", a, b) ("scaled", a, f) ("dpow", [exprs], scalar_expr)
def expr_evaluate(e, constant, common_poly, poly, challenge, negated, sum_, product, scaled):
    """`Expression::evaluate` (protocol.rs:310-373), same evaluation order."""
    ev = lambda x: expr_evaluate(x, constant, common_poly, poly, challenge, negated, sum_, product, scaled)
    tag = e[0]
    if tag == "const":
        return constant(e[1])
    if tag in ("identity", "lagrange"):
        return common_poly(e)
    if tag == "poly":
        return poly((e[1], e[2]))
    if tag == "challenge":
        return challenge(e[1])
    if tag == "neg":
        return negated(ev(e[1]))
    if tag == "sum":
        a = ev(e[1])
        b = ev(e[2])
        return sum_(a, b)
    if tag == "prod":
        a = ev(e[1])
        b = ev(e[2])
        return product(a, b)
    if tag == "scaled":
        return scaled(ev(e[1]), e[2])
    if tag == "dpow":
        exprs = e[1]
        assert exprs
        if len(exprs) == 1:
            return ev(exprs[0])
        acc = ev(exprs[0])
        scalar = ev(e[2])
        for x in exprs[1:]:
            acc = sum_(product(acc, scalar), ev(x))
        return acc
    raise ValueError(tag)


def _merge(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return a | b


def used_lagrange(e):
    r = expr_evaluate(e, lambda _: None, lambda p: {p[1]} if p[0] == "lagrange" else None, lambda _: None,
                      lambda _: None, lambda a: a, _merge, _merge, lambda a, _: a)
    return r or set()


def used_query(e):
    r = expr_evaluate(e, lambda _: None, lambda _: None, lambda q: {q}, lambda _: None, lambda a: a, _merge, _merge,
                      lambda a, _: a)
    return r or set()


# ---------------------------------------------------------------- protocol
def protocol_lagranges(pr):
    """`PlonkProtocol::langranges` (protocol.rs:80-111)."""
    out = set(used_lagrange(pr["quotient"]["numerator"]))
    if pr.get("instance_committing_key") is None:
        off = len(pr["preprocessed"])
        qs = [q for q in used_query(pr["quotient"]["numerator"]) if off <= q[0] < off + len(pr["num_instance"])]
        mn = mx = 0
        for _, rot in sorted(qs):
            if rot < mn:
                mn = rot
            elif rot > mx:
                mx = rot
        max_len = max(pr["num_instance"]) if pr["num_instance"] else 0
        out |= set(range(-mx, max_len + abs(mn)))
    return out


class CommonPolyEval:
    """protocol.rs:196-272 (fractions evaluated immediately: the native loader's batch_invert)."""

    def __init__(self, domain, lagranges, z):
        self.zn = pow(z, domain.n, R)
        self.zn_minus_one = (self.zn - 1) % R
        self.zn_minus_one_inv = pow(self.zn_minus_one, -1, R)
        numer = self.zn_minus_one * domain.n_inv % R
        self.identity = z
        self.lagrange = {}
        for i in sorted(set(lagranges)):
            omega = domain.rotate_scalar(1, i)
            self.lagrange[i] = numer * omega % R * pow((z - omega) % R, -1, R) % R

    def get(self, p):
        return self.identity if p[0] == "identity" else self.lagrange[p[1]]


class ProtocolError(Exception):
    """Error::InvalidProtocol / Error::InvalidInstances"""


def empty_queries(pr):  # proof.rs:170-181
    return [(poly, pr["domain"].rotate_scalar(1, rot)) for poly, rot in pr["queries"]]


def _query_set_count(pr, mos):
    if mos == "gwc19":
        return len({shift for _, shift in empty_queries(pr)})
    raise ValueError


def read_pcs_proof(pr, t, mos):
    if mos == "gwc19":  # gwc19.rs:102-110
        v = t.squeeze_challenge()
        ws = [t.read_ec_point() for _ in range(_query_set_count(pr, mos))]
        u = t.squeeze_challenge()
        return {"v": v, "ws": ws, "u": u}
    if mos == "bgh19":  # pcs/ipa/multiopen/bgh19.rs:113-153 (svk.domain = the protocol's domain)
        import ipa as I

        return I.bgh19_read_proof(pr["domain"].k, [(poly, shift, 0) for poly, shift in empty_queries(pr)], t)
    mu = t.squeeze_challenge()  # bdfg21.rs:95-117
    gamma = t.squeeze_challenge()
    w = t.read_ec_point()
    z_prime = t.squeeze_challenge()
    w_prime = t.read_ec_point()
    return {"mu": mu, "gamma": gamma, "w": w, "z_prime": z_prime, "w_prime": w_prime}


def plonk_proof_read(pr, instances, t, mos):
    """`PlonkProof::read` (proof.rs:52-168)."""
    if pr.get("transcript_initial_state") is not None:
        t.common_scalar(pr["transcript_initial_state"])
    if pr["num_instance"] != [len(x) for x in instances]:
        raise ProtocolError("InvalidInstances")
    ick = pr.get("instance_committing_key")
    committed = None
    if ick is not None:
        committed = []
        for inst in instances:
            m = Msm.sum([Msm.base(b) * s for s, b in zip(inst, ick["bases"])]
                        + ([Msm.base(ick["constant"])] if ick.get("constant") is not None else []))
            committed.append(m.evaluate(None))
        for c in committed:
            t.common_ec_point(c)
    else:
        for inst in instances:
            for x in inst:
                t.common_scalar(x)
    witnesses, challenges = [], []
    for n, m in zip(pr["num_witness"], pr["num_challenge"]):
        witnesses += [t.read_ec_point() for _ in range(n)]
        challenges += [t.squeeze_challenge() for _ in range(m)]
    quotients = [t.read_ec_point() for _ in range(pr["quotient"]["num_chunk"])]
    z = t.squeeze_challenge()
    evaluations = [t.read_scalar() for _ in pr["evaluations"]]
    pcs = read_pcs_proof(pr, t, mos)
    old = []
    for idx in pr["accumulator_indices"]:
        old.append(K.limbs_from_repr([instances[i][j] for i, j in idx]))
    return {"committed_instances": committed, "witnesses": witnesses, "challenges": challenges, "quotients": quotients,
            "z": z, "evaluations": evaluations, "pcs": pcs, "old_accumulators": old}


def proof_evaluations(pr, instances, proof, cpe):  # proof.rs:299-349
    evals = {}
    if pr.get("instance_committing_key") is None:
        off = len(pr["preprocessed"])
        for q in sorted(used_query(pr["quotient"]["numerator"])):
            if off <= q[0] < off + len(pr["num_instance"]):
                inst = instances[q[0] - off]
                evals[q] = sum(x * cpe.get(("lagrange", i - q[1])) for i, x in enumerate(inst)) % R
    for q, e in zip(pr["evaluations"], proof["evaluations"]):
        evals[tuple(q)] = e
    return evals


def _msm_size(m):
    return len(m.bases)


def _msm_const(m):
    """`try_into_constant` (msm.rs:68-78): Some only for a base-free Msm; its value is the constant (or 0)."""
    if m.bases:
        return None
    return m.constant or 0


def proof_commitments(pr, proof, cpe, evals):  # proof.rs:199-297
    commitments = [Msm.base(p) for p in pr["preprocessed"]]
    if proof["committed_instances"] is not None:
        commitments += [Msm.base(p) for p in proof["committed_instances"]]
    else:
        commitments += [Msm() for _ in pr["num_instance"]]
    commitments += [Msm.base(p) for p in proof["witnesses"]]

    def poly(q):
        if q in evals:
            return Msm.const(evals[q])
        if q[1] == 0 and q[0] < len(commitments):
            return commitments[q[0]].copy()
        raise ProtocolError("Missing query %r" % (q,))

    def challenge(i):
        if i >= len(proof["challenges"]):
            raise ProtocolError("Missing challenge %d" % i)
        return Msm.const(proof["challenges"][i])

    def product(a, b):
        if _msm_size(a) == 0:
            return b * _msm_const(a)
        if _msm_size(b) == 0:
            return a * _msm_const(b)
        raise ProtocolError("Invalid linearization")

    numerator = expr_evaluate(pr["quotient"]["numerator"], lambda s: Msm.const(s), lambda p: Msm.const(cpe.get(p)),
                              poly, challenge, lambda a: -a, lambda a, b: a + b, product, lambda a, s: a * s)
    quotient_query = (len(pr["preprocessed"]) + len(pr["num_instance"]) + len(proof["witnesses"]), 0)
    coeffs = K.powers(pow(cpe.zn, pr["quotient"]["chunk_degree"], R), len(proof["quotients"]))
    quotient = Msm.sum(Msm.base(ch) * co for co, ch in zip(coeffs, proof["quotients"]))
    lin = pr.get("linearization")
    if lin == "WithoutConstant":
        lq = (quotient_query[0] + 1, 0)
        msm, constant = Msm(None, numerator.scalars, numerator.bases), numerator.constant
        commitments.append(quotient)
        commitments.append(msm)
        evals[quotient_query] = ((constant or 0) + evals[lq]) * cpe.zn_minus_one_inv % R
    elif lin == "MinusVanishingTimesQuotient":
        t = numerator - quotient * cpe.zn_minus_one
        commitments.append(Msm(None, t.scalars, t.bases))
        evals[quotient_query] = t.constant or 0
    else:
        commitments.append(quotient)
        c = _msm_const(numerator)
        if c is None:
            raise ProtocolError("Invalid linearization")
        evals[quotient_query] = c * cpe.zn_minus_one_inv % R
    return commitments


def proof_queries(pr, evals):  # proof.rs:183-197
    return [(poly, shift, evals[(p, r)]) for (poly, shift), (p, r) in zip(empty_queries(pr), pr["queries"])]


def succinct_verify_msms(g, pr, instances, proof, mos):
    """`PlonkSuccinctVerifier::verify` (plonk.rs:58-92) up to the two Msm of the PCS."""
    cpe = CommonPolyEval(pr["domain"], protocol_lagranges(pr), proof["z"])
    evals = proof_evaluations(pr, instances, proof, cpe)
    commitments = proof_commitments(pr, proof, cpe, evals)
    queries = proof_queries(pr, evals)
    p = proof["pcs"]
    if mos == "gwc19":
        return K.gwc19_msms(g, commitments, proof["z"], queries, p["v"], p["ws"], p["u"])
    return K.bdfg21_msms(g, commitments, proof["z"], queries, p["mu"], p["gamma"], p["w"], p["z_prime"], p["w_prime"])


def succinct_verify(g, pr, instances, proof, mos):
    lhs, rhs = succinct_verify_msms(g, pr, instances, proof, mos)
    return [(lhs.evaluate(g), rhs.evaluate(g))] + list(proof["old_accumulators"])


def succinct_verify_ipa(g0, h, s, pr, instances, proof):
    """`PlonkSuccinctVerifier<IpaAs<Bgh19>>::verify` (plonk.rs:58-92 with the PCS of bgh19.rs:48-96):
    -> [IpaAccumulator] (no old accumulators: the IPA side has no `AccumulatorEncoding`, pcs.rs:173-184)."""
    import ipa as I

    cpe = CommonPolyEval(pr["domain"], protocol_lagranges(pr), proof["z"])
    evals = proof_evaluations(pr, instances, proof, cpe)
    commitments = proof_commitments(pr, proof, cpe, evals)
    queries = proof_queries(pr, evals)
    return [I.bgh19_verify(g0, h, s, commitments, proof["z"], queries, proof["pcs"])]


# ---------------------------------------------------------------- forger (toy SRS)
class _Dlog:
    """Points with known discrete logs: mint(c) = [c]G, remembered."""

    def __init__(self):
        self.table = {}

    def mint(self, c):
        c %= R
        p = O.g1_mul(O.G1_GEN, c) if c else None
        self.table[p] = c
        return p

    def of_msm(self, m):
        acc = m.constant or 0  # the constant multiplies g = G
        for s, b in zip(m.scalars, m.bases):
            acc += s * self.table[b]
        return acc % R


def _forge_front(pr, instances, make_transcript, rng, preprocessed_dlogs):
    """Everything up to the PCS opening: random commitments with known discrete logs, random
    evaluations.  -> (transcript, dlog table, z, commitments, queries)"""
    d = _Dlog()
    for p, c in zip(pr["preprocessed"], preprocessed_dlogs):
        d.table[p] = c % R
    d.table[O.G1_GEN] = 1
    d.table.update(pr.get("_known_dlogs", {}))
    t = make_transcript()
    if pr.get("transcript_initial_state") is not None:
        t.common_scalar(pr["transcript_initial_state"])
    ick = pr.get("instance_committing_key")
    committed = None
    if ick is not None:
        committed = []
        for inst in instances:
            m = Msm.sum([Msm.base(b) * s for s, b in zip(inst, ick["bases"])]
                        + ([Msm.base(ick["constant"])] if ick.get("constant") is not None else []))
            committed.append(m.evaluate(None))
        for c in committed:
            t.common_ec_point(c)
    else:
        for inst in instances:
            for x in inst:
                t.common_scalar(x)
    witnesses, challenges = [], []
    for n, m in zip(pr["num_witness"], pr["num_challenge"]):
        for _ in range(n):
            p = d.mint(rng.randrange(1, R))
            t.write_ec_point(p)
            witnesses.append(p)
        challenges += [t.squeeze_challenge() for _ in range(m)]
    quotients = []
    for _ in range(pr["quotient"]["num_chunk"]):
        p = d.mint(rng.randrange(1, R))
        t.write_ec_point(p)
        quotients.append(p)
    z = t.squeeze_challenge()
    evaluations = [rng.randrange(R) for _ in pr["evaluations"]]
    for e in evaluations:
        t.write_scalar(e)
    proof = {"committed_instances": committed, "witnesses": witnesses, "challenges": challenges,
             "quotients": quotients, "z": z, "evaluations": evaluations, "old_accumulators": []}
    if committed is not None:
        for p, inst in zip(committed, instances):  # dlogs of the instance commitments
            d.table[p] = d.of_msm(Msm.sum([Msm.base(b) * s for s, b in zip(inst, ick["bases"])]
                                          + ([Msm.base(ick["constant"])] if ick.get("constant") is not None else [])))
    cpe = CommonPolyEval(pr["domain"], protocol_lagranges(pr), z)
    evals = proof_evaluations(pr, instances, proof, cpe)
    commitments = proof_commitments(pr, proof, cpe, evals)
    queries = proof_queries(pr, evals)
    return t, d, z, commitments, queries


def forge_proof(pr, instances, secret, make_transcript, mos, rng, preprocessed_dlogs):
    """Writes a proof that verifies under the toy SRS ([1]G, [s]G2).
    `preprocessed_dlogs[i]` is the discrete log of pr["preprocessed"][i].
    Returns the proof bytes."""
    t, d, z, commitments, queries = _forge_front(pr, instances, make_transcript, rng, preprocessed_dlogs)
    if mos == "gwc19":
        v = t.squeeze_challenge()
        sets = K.gwc19_query_sets(queries)
        pv = K.powers(v, max(len(st["polys"]) for st in sets))
        for st in sets:
            f = sum(pvi * (d.of_msm(commitments[p]) - e) for p, e, pvi in zip(st["polys"], st["evals"], pv)) % R
            w = f * pow((secret - st["shift"] * z) % R, -1, R) % R
            t.write_ec_point(d.mint(w))
        t.squeeze_challenge()
    else:
        mu = t.squeeze_challenge()
        gamma = t.squeeze_challenge()
        w = d.mint(rng.randrange(1, R))
        t.write_ec_point(w)
        z_prime = t.squeeze_challenge()
        placeholder = ("W'",)  # symbolic base: solve dlog(lhs) = s * w'
        lhs, _ = K.bdfg21_msms(O.G1_GEN, commitments, z, queries, mu, gamma, w, z_prime, placeholder)
        rest = Msm(lhs.constant, [], [])
        coeff = 0
        for s, b in zip(lhs.scalars, lhs.bases):
            if b == placeholder:
                coeff = (coeff + s) % R
            else:
                rest.push(s, b)
        wp = d.of_msm(rest) * pow((secret - coeff) % R, -1, R) % R
        t.write_ec_point(d.mint(wp))
    return t.finalize()


def forge_proof_ipa(pr, instances, key_dlogs, make_transcript, rng, preprocessed_dlogs):
    """A PLONK proof over `IpaAs<Bgh19>` that verifies AND decides under a toy committing key whose
    discrete logs are known: key_dlogs = {"g": [gamma_i] (2^k of them), "h": eta, "s": sigma}.
    Commitments and evaluations are random; the multi-open part is random too, and the last scalar
    `c` of the IPA opening is solved from  dlog(C_k) = c (dlog U + h_eval xi_0 eta)  with
    U = <h_coeffs(xi), G> the honest value, so `decide` holds as well.  Returns the proof bytes."""
    import ipa as I

    k = pr["domain"].k
    assert len(key_dlogs["g"]) == 1 << k
    t, d, z, commitments, queries = _forge_front(pr, instances, make_transcript, rng, preprocessed_dlogs)
    g = [d.mint(c) for c in key_dlogs["g"]]
    s_pt = d.mint(key_dlogs["s"])
    d.mint(key_dlogs["h"])
    x_1 = t.squeeze_challenge()
    x_2 = t.squeeze_challenge()
    f = d.mint(rng.randrange(1, R))
    t.write_ec_point(f)
    x_3 = t.squeeze_challenge()
    q_evals = [rng.randrange(R) for _ in K.bdfg21_query_sets(queries)]
    for q in q_evals:
        t.write_scalar(q)
    x_4 = t.squeeze_challenge()
    p = I.bgh19_final_msm(g[0], commitments, z, queries, dict(x_1=x_1, x_2=x_2, f=f, x_3=x_3, q_evals=q_evals, x_4=x_4))
    c_bar = d.mint(rng.randrange(1, R))
    t.write_ec_point(c_bar)
    alpha = t.squeeze_challenge()
    xi_0 = t.squeeze_challenge()
    omega_prime = rng.randrange(R)
    lhs = (d.of_msm(p) + alpha * d.table[c_bar] - omega_prime * d.table[s_pt]) % R
    xi = []
    for _ in range(k):
        l, r = d.mint(rng.randrange(1, R)), d.mint(rng.randrange(1, R))
        t.write_ec_point(l)
        t.write_ec_point(r)
        x = t.squeeze_challenge()
        lhs = (lhs + pow(x, -1, R) * d.table[l] + x * d.table[r]) % R
        xi.append(x)
    u_dlog = sum(hc * gm for hc, gm in zip(I.h_coeffs(xi, 1), key_dlogs["g"])) % R
    c = lhs * pow((u_dlog + I.h_eval(xi, x_3) * xi_0 % R * key_dlogs["h"]) % R, -1, R) % R
    t.write_scalar(c)
    t.write_scalar(omega_prime)
    t.write_ec_point(d.mint(u_dlog))
    return t.finalize()
