#!/usr/bin/env python3
"""Generates bn254_consts.h and pallas_consts.h (both committed) from the curve definitions.

Self-contained on purpose: product constants must not depend on `oracle/`.
Run:  python gen_consts.py > bn254_consts.h ; python gen_consts.py pallas > pallas_consts.h

Two layers of names:
  SNARKV_FQ_* / SNARKV_FQ29_* / SNARKV_FR29_* / SNARKV_G1_B_MONT  -- what the field and group layers
      (fq.h, fq29.h, fr29.h, g1.h) read: base field p, scalar field r, curve constant b of
      y^2 = x^3 + b.  Emitted for every curve.
  BN254_*  -- everything only BN254 has: the GLV lattice, the tower, the pairing (bn254_consts.h only).
"""
import sys

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
X = 4965661367192848881
MONT_R = 1 << 256


def generic_section(p, r, b, name):
    def lim(v):
        return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(8))

    def lim29(v):
        return ", ".join("0x%08x" % ((v >> (29 * i)) & ((1 << 29) - 1)) for i in range(9))

    r29 = 1 << 261
    print("// ---- field / group layer constants of %s: base field p, scalar field r, y^2 = x^3 + %d" % (name, b))
    print("#define SNARKV_CURVE_NAME \"%s\"" % name)
    print("#define SNARKV_FQ_P_LIMBS { %s }" % lim(p))
    print("#define SNARKV_FQ_P_INV32 0x%08xu  // -p^-1 mod 2^32" % ((-pow(p, -1, 1 << 32)) % (1 << 32)))
    print("#define SNARKV_FQ_ONE_MONT { %s }  // 2^256 mod p" % lim(MONT_R % p))
    print("#define SNARKV_FQ_R2_MONT { %s }  // 2^512 mod p" % lim(MONT_R * MONT_R % p))
    print("#define SNARKV_FQ_P_MINUS_2_LIMBS { %s }" % lim(p - 2))
    print("#define SNARKV_G1_B_MONT { %s }  // b * 2^256 mod p" % lim(b * MONT_R % p))
    print("#define SNARKV_FR_R_LIMBS { %s }" % lim(r))
    print("#define SNARKV_FQ29_P_LIMBS { %s }" % lim29(p))
    print("#define SNARKV_FQ29_NINV 0x%08x  // -p^-1 mod 2^29" % ((-pow(p, -1, 1 << 29)) % (1 << 29)))
    print("#define SNARKV_FQ29_ONE_LIMBS { %s }  // 2^261 mod p" % lim29(r29 % p))
    print("#define SNARKV_FQ29_R2_LIMBS { %s }  // 2^522 mod p" % lim29(r29 * r29 % p))
    # halo2curves / pasta_curves keep field elements as a * 2^256 mod p in 4 x u64: in by 2^266 (-> a * 2^261), out by 2^256
    print("#define SNARKV_FQ29_M256_IN_LIMBS { %s }  // 2^266 mod p" % lim29((1 << 266) % p))
    print("#define SNARKV_FQ29_M256_OUT_LIMBS { %s }  // 2^256 mod p" % lim29(MONT_R % p))
    print("#define SNARKV_FR29_P_LIMBS { %s }  // r" % lim29(r))
    print("#define SNARKV_FR29_NINV 0x%08x  // -r^-1 mod 2^29" % ((-pow(r, -1, 1 << 29)) % (1 << 29)))
    print("#define SNARKV_FR29_ONE_LIMBS { %s }  // 2^261 mod r" % lim29(r29 % r))
    print("#define SNARKV_FR29_R2_LIMBS { %s }  // 2^522 mod r" % lim29(r29 * r29 % r))


def _cube_roots(m):
    for g_ in range(2, 50):
        w = pow(g_, (m - 1) // 3, m)
        if w != 1:
            return w, w * w % m


def _affine_mul(pt, k, p):
    def add(a, b):
        if a is None:
            return b
        if b is None:
            return a
        if a[0] == b[0]:
            if (a[1] + b[1]) % p == 0:
                return None
            l = 3 * a[0] * a[0] * pow(2 * a[1], -1, p) % p
        else:
            l = (b[1] - a[1]) * pow(b[0] - a[0], -1, p) % p
        x = (l * l - a[0] - b[0]) % p
        return (x, (l * (a[0] - x) - a[1]) % p)

    acc = None
    for bit in bin(k)[2:]:
        acc = add(acc, acc)
        if bit == "1":
            acc = add(acc, pt)
    return acc


def _glv_lattice(n, lam):
    import math

    rs, ts = [n, lam], [0, 1]
    sq = math.isqrt(n)
    while rs[-1] >= sq:
        q = rs[-2] // rs[-1]
        rs.append(rs[-2] - q * rs[-1])
        ts.append(ts[-2] - q * ts[-1])
    q = rs[-2] // rs[-1]
    cand = [(rs[-2], -ts[-2]), (rs[-2] - q * rs[-1], -(ts[-2] - q * ts[-1]))]
    a2, b2 = min(cand, key=lambda v: v[0] * v[0] + v[1] * v[1])
    return rs[-1], -ts[-1], a2, b2


def glv_section(p, r, gen):
    """GLV endomorphism phi(x, y) = (beta x, y) = lambda (x, y) of a j = 0 curve (glv.h): beta, the
    reduced lattice (a1, b1), (a2, b2) of {(a, b): a + b lambda = 0 mod r} with b1 < 0 < a1, a2, b2, and
    g_i = floor(2^288 |b_j| / r).  Uniform widths: lattice entries 4 words (<= 128 bits), g_i 6 words.
    The 288-bit shift keeps the rounding error of c_i = round(k b / r) below 2^-34, so the remainder is
    (alpha, beta) in (-1/2 - 2^-34, 1/2 + 2^-34)^2 of the lattice basis and |k_i| <= (1/2 + 2^-34) (|x1| + |x2|)."""
    def words(v, n):
        assert 0 <= v < 1 << (32 * n)
        return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n))

    def lim29(v):
        return ", ".join("0x%08x" % ((v >> (29 * i)) & ((1 << 29) - 1)) for i in range(9))

    lam = _cube_roots(r)[0]
    lg = _affine_mul(gen, lam, p)
    beta = [b for b in _cube_roots(p) if (b * gen[0] % p, gen[1]) == lg][0]
    a1, b1, a2, b2 = _glv_lattice(r, lam)
    assert a1 * b2 - a2 * b1 == r and b1 < 0 < a1 and a2 > 0 and b2 > 0
    assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0
    # |k1| <= (1/2 + 2^-34)(a1 + a2), |k2| <= (1/2 + 2^-34)(|b1| + b2): both must stay below 2^127 (bit 127 = sign)
    assert a1 + a2 < (1 << 128) - (1 << 120) and -b1 + b2 < (1 << 128) - (1 << 120)
    g1, g2 = (b2 << 288) // r, ((-b1) << 288) // r
    print("// GLV: lambda^2 + lambda + 1 = 0 mod r, phi(x, y) = (beta x, y) = lambda (x, y);")
    print("// lattice (a1, b1), (a2, b2) with a + b lambda = 0 mod r, b1 < 0; g_i = floor(2^288 |b_j| / r)")
    print("// lambda = 0x%x" % lam)
    print("#define SNARKV_GLV_BETA29_LIMBS { %s }  // beta * 2^261 mod p" % lim29(beta * (1 << 261) % p))
    print("#define SNARKV_GLV_A1 { %s }  // %d bits" % (words(a1, 4), a1.bit_length()))
    print("#define SNARKV_GLV_NEG_B1 { %s }  // |b1|, %d bits" % (words(-b1, 4), (-b1).bit_length()))
    print("#define SNARKV_GLV_A2 { %s }  // %d bits" % (words(a2, 4), a2.bit_length()))
    print("#define SNARKV_GLV_B2 { %s }  // %d bits" % (words(b2, 4), b2.bit_length()))
    print("#define SNARKV_GLV_G1 { %s }  // floor(2^288 b2 / r), %d bits" % (words(g1, 6), g1.bit_length()))
    print("#define SNARKV_GLV_G2 { %s }  // floor(2^288 |b1| / r), %d bits" % (words(g2, 6), g2.bit_length()))


if len(sys.argv) > 1 and sys.argv[1] == "pallas":
    # pallas (pasta): y^2 = x^3 + 5 over Fp, group order q (= the base field of vesta); generator (-1, 2).
    PP = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
    PQ = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
    assert PP == (1 << 254) + 45560315531419706090280762371685220353
    assert PQ == (1 << 254) + 45560315531506369815346746415080538113
    assert (2 * 2 - ((-1) ** 3 + 5)) % PP == 0  # the generator is on the curve
    print("// GENERATED by gen_consts.py pallas -- do not edit.  8x32-bit limbs little-endian; *_MONT in Montgomery")
    print("// form (R = 2^256); *29* in 9x29-bit limbs (R = 2^261).  No pairing constants: the pasta build has no KZG decider.")
    print("#pragma once")
    generic_section(PP, PQ, 5, "pallas")
    glv_section(PP, PQ, (PP - 1, 2))
    sys.exit(0)


def limbs(v):
    return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(8))


def mont(v):
    return v * MONT_R % P


def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2mul(r, a)
        a = f2mul(a, a)
        e >>= 1
    return r


def f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, -a[1] * d % P)


XI = (9, 1)
print("// GENERATED by gen_consts.py -- do not edit.  BN254 constants, 8x32-bit limbs,")
print("// little-endian limb order; *_MONT values are in Montgomery form (R = 2^256).")
print("#pragma once")
generic_section(P, R, 3, "bn254")
glv_section(P, R, (1, 2))
print("// ---- BN254 only")
print("#define BN254_P_LIMBS { %s }" % limbs(P))
print("#define BN254_R_LIMBS { %s }" % limbs(R))
print("#define BN254_P_INV32 0x%08xu  // -p^-1 mod 2^32" % ((-pow(P, -1, 1 << 32)) % (1 << 32)))
print("#define BN254_ONE_MONT { %s }  // R mod p" % limbs(mont(1)))
print("#define BN254_R2_MONT { %s }  // R^2 mod p" % limbs(MONT_R * MONT_R % P))
print("#define BN254_THREE_MONT { %s }" % limbs(mont(3)))
b2 = f2mul((3, 0), f2inv(XI))
print("#define BN254_TWIST_B_C0_MONT { %s }  // 3/(9+u)" % limbs(mont(b2[0])))
print("#define BN254_TWIST_B_C1_MONT { %s }" % limbs(mont(b2[1])))
b2_3 = f2mul((3, 0), b2)
print("#define BN254_TWIST_3B_C0_MONT { %s }  // 9/(9+u)" % limbs(mont(b2_3[0])))
print("#define BN254_TWIST_3B_C1_MONT { %s }" % limbs(mont(b2_3[1])))
print("#define BN254_P_MINUS_2_LIMBS { %s }" % limbs(P - 2))
print("#define BN254_P_PLUS_1_DIV_4_LIMBS { %s }" % limbs((P + 1) // 4))
print("#define BN254_X_U64 0x%016xull" % X)
print("#define BN254_ATE_LOOP_LO 0x%016xull  // (6x+2) low 64 bits" % ((6 * X + 2) & (2**64 - 1)))
print("#define BN254_ATE_LOOP_HI 0x%xull  // (6x+2) >> 64" % ((6 * X + 2) >> 64))
# radix-2^29 signed-limb form (fq29.h), Montgomery R = 2^261
def limbs29(v):
    return ", ".join("0x%08x" % ((v >> (29 * i)) & ((1 << 29) - 1)) for i in range(9))


R29 = 1 << 261
print("#define BN254_P29_LIMBS { %s }" % limbs29(P))
print("#define BN254_P29_NINV 0x%08x  // -p^-1 mod 2^29" % ((-pow(P, -1, 1 << 29)) % (1 << 29)))
print("#define BN254_ONE29_LIMBS { %s }  // 2^261 mod p" % limbs29(R29 % P))
print("#define BN254_R2_29_LIMBS { %s }  // 2^522 mod p" % limbs29(R29 * R29 % P))
print("#define BN254_THREE29_LIMBS { %s }  // 3 * 2^261 mod p" % limbs29(3 * R29 % P))
# the scalar field in the same form (fr29.h: Poseidon transcripts on the device)
print("#define BN254_FR29_LIMBS { %s }  // r" % limbs29(R))
print("#define BN254_FR29_NINV 0x%08x  // -r^-1 mod 2^29" % ((-pow(R, -1, 1 << 29)) % (1 << 29)))
print("#define BN254_FR29_ONE_LIMBS { %s }  // 2^261 mod r" % limbs29(R29 % R))
print("#define BN254_FR29_R2_LIMBS { %s }  // 2^522 mod r" % limbs29(R29 * R29 % R))
# Frobenius coefficients gamma_{k,i} = xi^(i (p^k - 1)/6), k=1..3, i=1..5
for k in (1, 2, 3):
    print("// gamma_%d,i = xi^(i*(p^%d-1)/6), i = 1..5 : {c0, c1} Montgomery" % (k, k))
    print("#define BN254_FROB_GAMMA_%d { \\" % k)
    for i in range(1, 6):
        g = f2pow(XI, i * (P**k - 1) // 6)
        print("  { { %s }, { %s } }, \\" % (limbs(mont(g[0])), limbs(mont(g[1]))))
    print("}")
# the same Frobenius coefficients in the 9x29-bit Montgomery (R = 2^261) form
for k in (1, 2, 3):
    print("#define BN254_FROB29_GAMMA_%d { \\" % k)
    for i in range(1, 6):
        gg = f2pow(XI, i * (P**k - 1) // 6)
        print("  { { %s }, { %s } }, \\" % (limbs29(gg[0] * R29 % P), limbs29(gg[1] * R29 % P)))
    print("}")
# twist Frobenius constants (for Q1 = pi(Q), Q2 = pi^2(Q))
for name, e in (("G12", (P - 1) // 3), ("G13", (P - 1) // 2), ("G22", (P * P - 1) // 3), ("G23", (P * P - 1) // 2)):
    g = f2pow(XI, e)
    print("#define BN254_TWIST_%s { { %s }, { %s } }" % (name, limbs(mont(g[0])), limbs(mont(g[1]))))
