// Optimal-ate pairing pieces for the KZG decider
//   accept  <=>  e(lhs, g2) * e(rhs, -s_g2) == 1
// (reference `snark-verifier/src/pcs/kzg/decider.rs:70-82`: G2Prepared::from x2,
// `multi_miller_loop`, `final_exponentiation`, `is_identity`).
//
// Design: both G2 points are constants of the deciding key, yet the reference
// rebuilds `G2Prepared` on every `decide` (decider.rs:74).  Here the line
// coefficients are computed ONCE per deciding key (`g2_prepare`, projective,
// inversion-free) and kept in HBM; the Miller loop proper then contains no G2
// arithmetic at all -- only Fq12 squarings and sparse line products.
#pragma once
#include "tower.h"

namespace snarkv {

// 6x+2 has 65 bits, 37 of them set: 64 doubling lines + 36 addition lines
// + 2 Frobenius lines.
constexpr int kAteBits = 65;
constexpr int kLinesPerG2 = 64 + 36 + 2;

struct G2Affine {
  Fq2 x, y;
};

// line  l(P) = cy * yP  +  cx * xP * w  +  cw * w^3   (up to an Fq2 factor,
// which the final exponentiation kills since (p^2-1) | (p^12-1)/r).
struct LineCoeff {
  Fq2 cy, cx, cw;
};

struct G2Prepared {
  LineCoeff line[kLinesPerG2];
  uint32_t is_identity;
};

SNARKV_HD bool ate_bit(int i) {
  return i < 64 ? ((BN254_ATE_LOOP_LO >> i) & 1ull) != 0 : ((BN254_ATE_LOOP_HI >> (i - 64)) & 1ull) != 0;
}

struct G2Proj {
  Fq2 x, y, z;
};

// Tangent at T=(X,Y,Z) (x=X/Z, y=Y/Z on y^2 = x^3 + b'), scaled by 2YZ*Z/Z:
//   cy = 2YZ, cx = -3X^2, cw = Y^2 - 3b'Z^2       [uses Y^2 Z = X^3 + b' Z^3]
// then T <- 2T with  A=3X^2, B=2YZ, N=A^2 Z - 2 X B^2:
//   X3 = N B,  Y3 = A (X B^2 - N) - Y B^3,  Z3 = B^3 Z.
SNARKV_TW void g2_double_step(G2Proj& t, LineCoeff& l) {
  constexpr uint32_t b3c0[8] = BN254_TWIST_3B_C0_MONT;
  constexpr uint32_t b3c1[8] = BN254_TWIST_3B_C1_MONT;
  Fq2 b3;
  for (int i = 0; i < 8; ++i) {
    b3.c0.v[i] = b3c0[i];
    b3.c1.v[i] = b3c1[i];
  }
  Fq2 xx = fq2_sqr(t.x);
  Fq2 a = fq2_add(fq2_dbl(xx), xx);       // 3X^2
  Fq2 b = fq2_dbl(fq2_mul(t.y, t.z));     // 2YZ
  Fq2 yy = fq2_sqr(t.y);
  Fq2 zz = fq2_sqr(t.z);
  l.cy = b;
  l.cx = fq2_neg(a);
  l.cw = fq2_sub(yy, fq2_mul(b3, zz));
  Fq2 bb = fq2_sqr(b);
  Fq2 xbb = fq2_mul(t.x, bb);
  Fq2 n = fq2_sub(fq2_mul(fq2_sqr(a), t.z), fq2_dbl(xbb));
  Fq2 bbb = fq2_mul(bb, b);
  Fq2 y3 = fq2_sub(fq2_mul(a, fq2_sub(xbb, n)), fq2_mul(t.y, bbb));
  t.x = fq2_mul(n, b);
  t.y = y3;
  t.z = fq2_mul(bbb, t.z);
}

// Chord through T and affine Q=(x2,y2), with E = y2 Z - Y, F = x2 Z - X:
//   cy = F, cx = -E, cw = E x2 - F y2
// then T <- T+Q with D = F^2 Z, N = E^2 Z - X F^2 - x2 D:
//   X3 = N F, Y3 = E (x2 D - N) - y2 F D, Z3 = F D.
SNARKV_TW void g2_add_step(G2Proj& t, const G2Affine& q, LineCoeff& l) {
  Fq2 e = fq2_sub(fq2_mul(q.y, t.z), t.y);
  Fq2 f = fq2_sub(fq2_mul(q.x, t.z), t.x);
  l.cy = f;
  l.cx = fq2_neg(e);
  l.cw = fq2_sub(fq2_mul(e, q.x), fq2_mul(f, q.y));
  Fq2 ff = fq2_sqr(f);
  Fq2 d = fq2_mul(ff, t.z);
  Fq2 x2d = fq2_mul(q.x, d);
  Fq2 n = fq2_sub(fq2_sub(fq2_mul(fq2_sqr(e), t.z), fq2_mul(t.x, ff)), x2d);
  Fq2 fd = fq2_mul(f, d);
  Fq2 y3 = fq2_sub(fq2_mul(e, fq2_sub(x2d, n)), fq2_mul(q.y, fd));
  t.x = fq2_mul(n, f);
  t.y = y3;
  t.z = fd;
}

SNARKV_HD Fq2 fq2_const(const uint32_t (&c)[2][8]) {
  Fq2 r;
  for (int i = 0; i < 8; ++i) {
    r.c0.v[i] = c[0][i];
    r.c1.v[i] = c[1][i];
  }
  return r;
}

// Line table for one G2 point: the `G2Prepared::from` of the reference's
// decider, done once per deciding key.
SNARKV_TW void g2_prepare(const G2Affine& q, G2Prepared& out) {
  if (fq2_is_zero(q.x) && fq2_is_zero(q.y)) {
    out.is_identity = 1;
    return;
  }
  out.is_identity = 0;
  G2Proj t{q.x, q.y, fq2_one()};
  int idx = 0;
  for (int i = kAteBits - 2; i >= 0; --i) {
    g2_double_step(t, out.line[idx++]);
    if (ate_bit(i)) g2_add_step(t, q, out.line[idx++]);
  }
  constexpr uint32_t g12[2][8] = BN254_TWIST_G12;
  constexpr uint32_t g13[2][8] = BN254_TWIST_G13;
  constexpr uint32_t g22[2][8] = BN254_TWIST_G22;
  constexpr uint32_t g23[2][8] = BN254_TWIST_G23;
  G2Affine q1{fq2_mul(fq2_conj(q.x), fq2_const(g12)), fq2_mul(fq2_conj(q.y), fq2_const(g13))};
  G2Affine q2{fq2_mul(q.x, fq2_const(g22)), fq2_neg(fq2_mul(q.y, fq2_const(g23)))};  // -pi^2(Q)
  g2_add_step(t, q1, out.line[idx++]);
  g2_add_step(t, q2, out.line[idx++]);
}

struct G1AffineM {  // Montgomery affine G1 point as the Miller loop consumes it
  Fq x, y;
};

SNARKV_HD Fq12 line_mul(const Fq12& f, const LineCoeff& l, const G1AffineM& p) {
  return fq12_mul_by_line(f, fq2_mul_fq(l.cy, p.y), fq2_mul_fq(l.cx, p.x), l.cw);
}

// prod_k f_{6x+2,Q_k}(P_k) * Frobenius lines; squarings shared between pairs.
// Pairs with P = O or Q = O contribute 1 (as `multi_miller_loop` does).
SNARKV_TW Fq12 multi_miller_loop(const G1AffineM* ps, const G2Prepared* const* qs, int npairs) {
  Fq12 f = fq12_one();
  bool live[4];
  for (int k = 0; k < npairs; ++k)
    live[k] = !(fq_is_zero(ps[k].x) && fq_is_zero(ps[k].y)) && !qs[k]->is_identity;
  int idx = 0;
  for (int i = kAteBits - 2; i >= 0; --i) {
    f = fq12_sqr(f);
    for (int k = 0; k < npairs; ++k)
      if (live[k]) f = line_mul(f, qs[k]->line[idx], ps[k]);
    ++idx;
    if (ate_bit(i)) {
      for (int k = 0; k < npairs; ++k)
        if (live[k]) f = line_mul(f, qs[k]->line[idx], ps[k]);
      ++idx;
    }
  }
  for (int s = 0; s < 2; ++s) {
    for (int k = 0; k < npairs; ++k)
      if (live[k]) f = line_mul(f, qs[k]->line[idx], ps[k]);
    ++idx;
  }
  return f;
}

SNARKV_TW Fq12 fq12_exp_by_x(const Fq12& f) {
  Fq12 res = f;  // top bit (62) of x
  for (int i = 61; i >= 0; --i) {
    res = fq12_sqr(res);
    if ((BN254_X_U64 >> i) & 1ull) res = fq12_mul(res, f);
  }
  return res;
}

// f^((p^12-1)/r) exactly: easy part (p^6-1)(p^2+1), then the hard part
// (p^4-p^2+1)/r = l0 + l1 p + l2 p^2 + p^3 through the vectorial addition
// chain  y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36  (Devegili-Scott-Dahab).
SNARKV_TW Fq12 final_exponentiation(const Fq12& f_in) {
  Fq12 f = fq12_mul(fq12_conj(f_in), fq12_inv(f_in));  // f^(p^6-1)
  f = fq12_mul(fq12_frobenius(f, 2), f);                // ^(p^2+1)
  Fq12 fx = fq12_exp_by_x(f);
  Fq12 fx2 = fq12_exp_by_x(fx);
  Fq12 fx3 = fq12_exp_by_x(fx2);
  Fq12 y0 = fq12_mul(fq12_mul(fq12_frobenius(f, 1), fq12_frobenius(f, 2)), fq12_frobenius(f, 3));
  Fq12 y1 = fq12_conj(f);
  Fq12 y2 = fq12_frobenius(fx2, 2);
  Fq12 y3 = fq12_conj(fq12_frobenius(fx, 1));
  Fq12 y4 = fq12_conj(fq12_mul(fx, fq12_frobenius(fx2, 1)));
  Fq12 y5 = fq12_conj(fx2);
  Fq12 y6 = fq12_conj(fq12_mul(fx3, fq12_frobenius(fx3, 1)));
  Fq12 t0 = fq12_mul(fq12_mul(fq12_sqr(y6), y4), y5);
  Fq12 t1 = fq12_mul(fq12_mul(y3, y5), t0);
  t0 = fq12_mul(t0, y2);
  t1 = fq12_sqr(fq12_mul(fq12_sqr(t1), t0));
  t0 = fq12_mul(t1, y1);
  t1 = fq12_mul(t1, y0);
  return fq12_mul(fq12_sqr(t0), t1);
}

}  // namespace snarkv
