"""CPU: the curve-generic device headers (fq29.h, fr29.h, g1_29.h, glv.h) compiled for the HOST
with the BN254 and with the pallas constants (csrc/curve_consts.h) against the big-integer oracles: lazy
9x29-bit field products at the edges of the range, the XYZZ adders, the Jacobian doubling chain, and the
GLV split k = k1 + k2 lambda with |k_i| < 2^127 -- the packing the Pippenger relies on."""
import ctypes
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import bn254 as BN  # noqa: E402
import pallas as PA  # noqa: E402


def _lib(curve):
    d = os.path.join(ROOT, "tests", "hosttest")
    so = os.path.join(d, "libhosttest_%s.so" % curve)
    src = os.path.join(d, "hosttest_curve.cpp")
    csrc = os.path.join(ROOT, "snark-verifier_amd", "csrc")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)
                                            if f.endswith((".h", ".h"))])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        flags = ["-DSNARKV_CURVE_PALLAS"] if curve == "pallas" else []
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC"] + flags + ["-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    lib.hc_curve.restype = ctypes.c_char_p
    assert lib.hc_curve() == curve.encode()
    return lib


CURVES = {"bn254": BN, "pallas": PA}


def _points(curve, n):
    if curve == "pallas":
        return PA.sample_points(17, n)
    import coracle as C

    raw = C.sample_points(17, n)
    return [BN.g1_from_bytes(raw[64 * i:64 * i + 64]) for i in range(n)]


def _buf(n=64):
    return ctypes.create_string_buffer(n)


@pytest.mark.parametrize("curve", ["bn254", "pallas"])
def test_field_products(curve):
    L, O = _lib(curve), CURVES[curve]
    rnd = random.Random(1)
    P, R = O.P, O.R
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (1 << 253) % P, (1 << 254) % P, (1 << 232) - 1, 3]
    vals = edge + [rnd.randrange(P) for _ in range(40)]
    e = O.fe_to_bytes
    o = _buf(32)
    for a in vals:
        for b in (vals[:6] + vals[-6:]):
            L.hc_fq_mul(e(a), e(b), o)
            assert o.raw == e(a * b % P)
        L.hc_fq_sqr(e(a), o)
        assert o.raw == e(a * a % P)
        if a:
            L.hc_fq_inv(e(a), o)
            assert o.raw == e(pow(a, -1, P))
    for _ in range(50):
        a, b, c, d = (rnd.choice(vals) for _ in range(4))
        L.hc_fq_mul2(e(a), e(b), e(c), e(d), o)
        assert o.raw == e((a * b - c * d) % P)
    rv = [0, 1, R - 1, R - 2] + [rnd.randrange(R) for _ in range(30)]
    for a in rv:
        for b in rv[:8]:
            L.hc_fr_mul(e(a), e(b), o)
            assert o.raw == e(a * b % R)


@pytest.mark.parametrize("curve", ["bn254", "pallas"])
def test_group_adders_and_doubling_chain(curve):
    L, O = _lib(curve), CURVES[curve]
    pts = _points(curve, 12)
    b = O.g1_to_bytes
    o = _buf(64)
    for i in range(6):
        p, q = pts[i], pts[i + 1]
        L.hc_g1_add(b(p), b(q), o)
        assert o.raw == b(O.g1_add(p, q))
        L.hc_g1_add(b(p), b(p), o)          # P + P -> the doubling branch of the careful adder
        assert o.raw == b(O.g1_double(p))
        L.hc_g1_add(b(p), b(O.g1_neg(p)), o)  # P - P -> identity
        assert o.raw == bytes(64)
    acc = pts[0]
    for q in pts[1:]:
        acc = O.g1_add(acc, q)
    L.hc_g1_madd_chain(b(pts[0]), b"".join(b(q) for q in pts[1:]), len(pts) - 1, o)
    assert o.raw == b(acc)
    for n in (1, 2, 16, 113, 240):
        L.hc_g1_double_n(b(pts[3]), n, o)
        assert o.raw == b(O.g1_mul(pts[3], pow(2, n + 1, O.R)))


@pytest.mark.parametrize("curve", ["bn254", "pallas"])
def test_glv_split_is_exact_and_fits_127_bits(curve):
    L, O = _lib(curve), CURVES[curve]
    R = O.R
    pts = _points(curve, 3)
    o = _buf(64)
    L.hc_glv_phi(O.g1_to_bytes(pts[0]), o)
    phi = O.g1_from_bytes(o.raw)
    assert phi[1] == pts[0][1] and phi != pts[0] and O.g1_is_on_curve(phi)
    # lambda: the scalar phi acts as (found from the three cube roots of unity mod r)
    lam = None
    for g in range(2, 40):
        w = pow(g, (R - 1) // 3, R)
        if w != 1:
            lam = w if O.g1_mul(pts[0], w) == phi else w * w % R
            break
    assert O.g1_mul(pts[0], lam) == phi and (lam * lam + lam + 1) % R == 0
    rnd = random.Random(9)
    # scalars that push the rounding hardest sit near multiples of r / (lattice step); random + edges
    ks = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, (R + 1) // 2, lam, R - lam, (lam * 7) % R, 1 << 127, (1 << 128) - 1,
          (1 << 253), (1 << 254) % R] + [rnd.randrange(R) for _ in range(4000)]
    out = _buf(32)
    worst = 0
    for k in ks:
        L.hc_glv_decompose(O.fe_to_bytes(k), out)
        h1, h2 = int.from_bytes(out.raw[:16], "little"), int.from_bytes(out.raw[16:], "little")
        m1, m2 = h1 & ((1 << 127) - 1), h2 & ((1 << 127) - 1)
        k1 = -m1 if h1 >> 127 else m1
        k2 = -m2 if h2 >> 127 else m2
        assert (k1 + k2 * lam - k) % R == 0
        worst = max(worst, m1.bit_length(), m2.bit_length())
    assert worst <= 127
