"""Test helper: StandardPlonk-SHAPED protocols (the structure `compile` gives the
reference's example circuit, snark-verifier/src/system/halo2.rs:100-160,281-304,
729-744: 5 fixed + 3 permutation commitments, 1 instance column, witnesses in
phases [3 advice | 0 lookup | 2 perm-z + 1 random], challenges [theta | beta,
gamma | alpha], quotient in `degree-1` chunks, numerator = DistributePowers of
gate + permutation constraints) with random contents, for the forger in
oracle/plonk.py.  Also the byte packing the C++ test driver parses."""
import struct

import bn254 as O
import plonk as P

R = O.R


def use_curve(mod):
    """Synthesise protocols over another curve module (oracle/pallas.py); switches plonk.py / kzg.py / ipa.py too."""
    global O, R
    O, R = mod, mod.R
    P.use_curve(mod)


def const(v):
    return ("const", v % R)


def poly(p, rot=0):
    return ("poly", p, rot)


def add(*xs):
    acc = xs[0]
    for x in xs[1:]:
        acc = ("sum", acc, x)
    return acc


def mul(*xs):
    acc = xs[0]
    for x in xs[1:]:
        acc = ("prod", acc, x)
    return acc


def neg(x):
    return ("neg", x)


def standard_plonk_protocol(rng, k=6, num_instance=(2,), linearization=None, accumulator_rows=None,
                            committed_instances=False, initial_state=True, linearize_fixed=False):
    """returns (protocol, preprocessed_dlogs)"""
    dom = P.Domain(k)
    pre_dlogs = [rng.randrange(1, R) for _ in range(8)]
    preprocessed = [O.g1_mul(O.G1_GEN, c) for c in pre_dlogs]
    n_inst = len(num_instance)
    I0 = 8                       # first instance poly
    W = I0 + n_inst              # first witness poly
    a, b, c, z1, z2, rnd = W, W + 1, W + 2, W + 3, W + 4, W + 5
    last = -6
    q_a, q_b, q_c, q_ab, q_k, s1, s2, s3 = range(8)
    theta, beta, gamma, alpha = 0, 1, 2, 3
    ch = lambda i: ("challenge", i)
    l0, llast = ("lagrange", 0), ("lagrange", last)
    lblind = add(*[("lagrange", i) for i in range(last + 1, 0)])
    gate = add(mul(poly(q_a), poly(a)), mul(poly(q_b), poly(b)), mul(poly(q_c), poly(c)),
               mul(poly(q_ab), mul(poly(a), poly(b))), poly(q_k), poly(I0),
               ("scaled", mul(poly(a, 1), poly(c, -1)), rng.randrange(R)), mul(ch(theta), const(rng.randrange(R))))
    active = add(const(1), neg(add(llast, lblind)))
    delta = rng.randrange(2, R)
    lhs_p = mul(poly(z1, 1), add(poly(a), mul(ch(beta), poly(s1)), ch(gamma)), add(poly(b), mul(ch(beta), poly(s2)), ch(gamma)))
    rhs_p = mul(poly(z1), add(poly(a), mul(ch(beta), ("identity",)), ch(gamma)),
                add(poly(b), mul(ch(beta), ("scaled", ("identity",), delta)), ch(gamma)))
    constraints = [
        gate,
        mul(l0, add(const(1), neg(poly(z1)))),
        mul(llast, add(mul(poly(z2), poly(z2)), neg(poly(z2)))),
        mul(l0, add(poly(z2), neg(poly(z1, last)))),
        mul(active, add(lhs_p, neg(rhs_p))),
        mul(active, add(mul(poly(z2, 1), add(poly(c), mul(ch(beta), poly(s3)), ch(gamma))),
                        neg(mul(poly(z2), add(poly(c), ch(gamma)))))),
    ]
    if n_inst > 1:
        constraints.append(mul(poly(q_k), add(poly(I0 + 1), neg(poly(I0 + 1, 1)))))
    numerator = ("dpow", constraints, ch(alpha))
    evaluations = [(a, 0), (b, 0), (c, 0), (a, 1), (c, -1)] + [(i, 0) for i in range(5)] + [(rnd, 0)] + \
                  [(s1, 0), (s2, 0), (s3, 0)] + [(z1, 0), (z1, 1), (z1, last), (z2, 0), (z2, 1)]
    Q = W + 6                    # quotient poly index (proof.rs:243-246)
    lin_extra = []
    if linearization == "WithoutConstant":
        evaluations = evaluations + [(Q + 1, 0)]
        lin_extra = [(Q + 1, 0)]
    queries = [(a, 0), (b, 0), (c, 0), (a, 1), (c, -1), (z1, 0), (z1, 1), (z1, last), (z2, 0), (z2, 1)] + \
              [(i, 0) for i in range(5)] + [(s1, 0), (s2, 0), (s3, 0)] + [(Q, 0)] + lin_extra + [(rnd, 0)]
    if linearize_fixed:
        # q_k is not evaluated: with a linearization strategy its COMMITMENT enters the numerator Msm
        assert linearization is not None
        evaluations = [q for q in evaluations if q != (q_k, 0)]
        queries = [q for q in queries if q != (q_k, 0)]
    pr = {
        "domain": dom, "preprocessed": preprocessed, "num_instance": list(num_instance),
        "num_witness": [3, 0, 3], "num_challenge": [1, 2, 1], "evaluations": evaluations, "queries": queries,
        "quotient": {"chunk_degree": 1, "num_chunk": 3, "numerator": numerator},
        "transcript_initial_state": rng.randrange(R) if initial_state else None,
        "instance_committing_key": None, "linearization": linearization,
        "accumulator_indices": [[(0, j) for j in rows] for rows in (accumulator_rows or [])],
    }
    if committed_instances:
        bd = [rng.randrange(1, R) for _ in range(max(num_instance) + 1)]
        pts = [O.g1_mul(O.G1_GEN, c) for c in bd]
        pr["instance_committing_key"] = {"bases": pts[:-1], "constant": pts[-1]}
        pr["_known_dlogs"] = dict(zip(pts, bd))   # forger only (toy SRS); not part of the protocol
        pr["queries"] = [(I0 + t, 0) for t in range(n_inst)] + pr["queries"]
        pr["evaluations"] = [(I0 + t, 0) for t in range(n_inst)] + pr["evaluations"]
    return pr, pre_dlogs


# ---------------------------------------------------------------- packing for the C++ driver
def _fr(x):
    return O.fe_to_bytes(x % R)


def _u32(x):
    return struct.pack("<I", x)


def _i32(x):
    return struct.pack("<i", x)


def pack_expr(e):
    t = e[0]
    if t == "const":
        return b"\x00" + _fr(e[1])
    if t == "identity":
        return b"\x01"
    if t == "lagrange":
        return b"\x02" + _i32(e[1])
    if t == "poly":
        return b"\x03" + _u32(e[1]) + _i32(e[2])
    if t == "challenge":
        return b"\x04" + _u32(e[1])
    if t == "neg":
        return b"\x05" + pack_expr(e[1])
    if t == "sum":
        return b"\x06" + pack_expr(e[1]) + pack_expr(e[2])
    if t == "prod":
        return b"\x07" + pack_expr(e[1]) + pack_expr(e[2])
    if t == "scaled":
        return b"\x08" + pack_expr(e[1]) + _fr(e[2])
    if t == "dpow":
        return b"\x09" + _u32(len(e[1])) + b"".join(pack_expr(x) for x in e[1]) + pack_expr(e[2])
    raise ValueError(t)


def pack_protocol(pr):
    d = pr["domain"]
    out = _u32(d.k) + _fr(d.gen)
    out += _u32(len(pr["preprocessed"])) + b"".join(O.g1_to_bytes(p) for p in pr["preprocessed"])
    for key in ("num_instance", "num_witness", "num_challenge"):
        out += _u32(len(pr[key])) + b"".join(_u32(x) for x in pr[key])
    for key in ("evaluations", "queries"):
        out += _u32(len(pr[key])) + b"".join(_u32(p) + _i32(r) for p, r in pr[key])
    q = pr["quotient"]
    out += _u32(q["chunk_degree"]) + _u32(q["num_chunk"]) + pack_expr(q["numerator"])
    tis = pr.get("transcript_initial_state")
    out += b"\x01" + _fr(tis) if tis is not None else b"\x00"
    ick = pr.get("instance_committing_key")
    if ick is None:
        out += b"\x00"
    else:
        out += b"\x01" + _u32(len(ick["bases"])) + b"".join(O.g1_to_bytes(p) for p in ick["bases"])
        out += (b"\x01" + O.g1_to_bytes(ick["constant"])) if ick.get("constant") is not None else b"\x00"
    out += bytes([{None: 0, "WithoutConstant": 1, "MinusVanishingTimesQuotient": 2}[pr.get("linearization")]])
    out += _u32(len(pr["accumulator_indices"]))
    for idx in pr["accumulator_indices"]:
        out += _u32(len(idx)) + b"".join(_u32(i) + _u32(j) for i, j in idx)
    return out


def pack_instances(instances):
    return _u32(len(instances)) + b"".join(_u32(len(x)) + b"".join(_fr(v) for v in x) for x in instances)


# ---------------------------------------------------------------- random shapes (fuzzing the mirror against the oracle)
def random_protocol(rng, linearization=None):
    """A random but well-formed protocol: random counts of preprocessed / instance / witness
    polynomials over random phases, random rotations, a random expression tree over every node
    kind as numerator, random quotient chunking.  Every polynomial query the numerator makes has
    an evaluation (so it is a constant, as without linearization), except -- in linearized modes --
    a few rotation-0 queries that appear only linearly."""
    k = rng.randrange(3, 8)
    dom = P.Domain(k)
    n_pre = rng.randrange(1, 6)
    pre_dlogs = [rng.randrange(1, R) for _ in range(n_pre)]
    preprocessed = [O.g1_mul(O.G1_GEN, c) for c in pre_dlogs]
    num_instance = [rng.randrange(1, 4) for _ in range(rng.randrange(1, 3))]
    phases = rng.randrange(1, 4)
    num_witness = [rng.randrange(0, 4) for _ in range(phases)]
    if sum(num_witness) == 0:
        num_witness[0] = 1
    num_challenge = [rng.randrange(0, 3) for _ in range(phases)]
    if sum(num_challenge) == 0:
        num_challenge[-1] = 1
    n_ch = sum(num_challenge)
    I0 = n_pre
    W0 = I0 + len(num_instance)
    n_w = sum(num_witness)
    Q = W0 + n_w
    rots = [0, 1, -1, rng.randrange(-9, -2), 2]
    committed = [i for i in range(n_pre)] + [W0 + i for i in range(n_w)]
    qpool = [(p, rng.choice(rots)) for p in committed for _ in range(2)]
    qpool = list(dict.fromkeys(qpool))
    rng.shuffle(qpool)
    evaluated = qpool[:max(2, len(qpool) * 2 // 3)]
    inst_q = [(I0 + t, rng.choice([0, 0, 1, -1])) for t in range(len(num_instance))]
    linear_only = []
    if linearization is not None:
        cand = [(p, 0) for p in committed if (p, 0) not in evaluated]
        linear_only = cand[:2]

    def leaf():
        c = rng.randrange(7)
        if c == 0:
            return ("const", rng.randrange(R))
        if c == 1:
            return ("identity",)
        if c == 2:
            return ("lagrange", rng.choice([0, -1, 1, -4, 3]))
        if c == 3:
            return ("challenge", rng.randrange(n_ch))
        if c == 4:
            q = rng.choice(inst_q)
            return ("poly", q[0], q[1])
        q = rng.choice(evaluated)
        return ("poly", q[0], q[1])

    def tree(d):
        if d == 0 or rng.random() < 0.25:
            return leaf()
        c = rng.randrange(6)
        if c == 0:
            return ("neg", tree(d - 1))
        if c == 1:
            return ("sum", tree(d - 1), tree(d - 1))
        if c == 2:
            return ("prod", tree(d - 1), tree(d - 1))
        if c == 3:
            return ("scaled", tree(d - 1), rng.randrange(R))
        if c == 4:
            return ("dpow", [tree(d - 1) for _ in range(rng.randrange(1, 4))], leaf() if rng.random() < 0.5 else ("challenge", rng.randrange(n_ch)))
        return ("sum", tree(d - 1), leaf())

    numerator = ("dpow", [tree(3) for _ in range(rng.randrange(1, 5))], ("challenge", rng.randrange(n_ch)))
    for p, _ in linear_only:  # commitments entering linearly: (constant tree) * poly + ...
        numerator = ("sum", numerator, ("prod", ("scaled", ("challenge", rng.randrange(n_ch)), rng.randrange(R)), ("poly", p, 0)))
    evaluations = list(evaluated)
    num_chunk = rng.randrange(1, 5)
    queries = list(evaluated) + [(Q, 0)]
    if linearization == "WithoutConstant":
        evaluations.append((Q + 1, 0))
        queries.append((Q + 1, 0))
    rng.shuffle(queries)
    pr = {
        "domain": dom, "preprocessed": preprocessed, "num_instance": num_instance,
        "num_witness": num_witness, "num_challenge": num_challenge, "evaluations": evaluations, "queries": queries,
        "quotient": {"chunk_degree": rng.randrange(1, 4), "num_chunk": num_chunk, "numerator": numerator},
        "transcript_initial_state": rng.randrange(R) if rng.random() < 0.5 else None,
        "instance_committing_key": None, "linearization": linearization, "accumulator_indices": [],
    }
    return pr, pre_dlogs
