// BN254 extension tower for the pairing decider:
//   Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-xi) with xi = 9+u, Fq12 = Fq6[w]/(w^2-v)
// (SURVEY.md section 8a row A10 -- the tower halo2curves' bn256 uses; the
// reference reaches it only through `multi_miller_loop` /
// `final_exponentiation` at `snark-verifier/src/pcs/kzg/decider.rs:74-78`).
// Written from the field definitions; Karatsuba at every level.
#pragma once
#include "fq.h"

#if defined(__HIPCC__)
#define SNARKV_TW static __host__ __device__ __noinline__
#else
#define SNARKV_TW inline
#endif

namespace snarkv {

struct Fq2 {
  Fq c0, c1;
};
struct Fq6 {
  Fq2 c0, c1, c2;
};
struct Fq12 {
  Fq6 c0, c1;
};

// Out-of-line Fq product for the tower: the pairing is far too large to inline
// 400-instruction Montgomery products everywhere.
SNARKV_TW Fq fq_mul_ol(const Fq& a, const Fq& b) { return fq_mul(a, b); }

// ---------------------------------------------------------------- Fq2
SNARKV_HD Fq2 fq2_zero() { return Fq2{fq_zero(), fq_zero()}; }
SNARKV_HD Fq2 fq2_one() { return Fq2{fq_one(), fq_zero()}; }
SNARKV_HD bool fq2_is_zero(const Fq2& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
SNARKV_HD bool fq2_eq(const Fq2& a, const Fq2& b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
SNARKV_HD Fq2 fq2_add(const Fq2& a, const Fq2& b) { return Fq2{fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; }
SNARKV_HD Fq2 fq2_sub(const Fq2& a, const Fq2& b) { return Fq2{fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; }
SNARKV_HD Fq2 fq2_neg(const Fq2& a) { return Fq2{fq_neg(a.c0), fq_neg(a.c1)}; }
SNARKV_HD Fq2 fq2_dbl(const Fq2& a) { return Fq2{fq_dbl(a.c0), fq_dbl(a.c1)}; }
SNARKV_HD Fq2 fq2_conj(const Fq2& a) { return Fq2{a.c0, fq_neg(a.c1)}; }

SNARKV_TW Fq2 fq2_mul(const Fq2& a, const Fq2& b) {
  Fq t0 = fq_mul_ol(a.c0, b.c0);
  Fq t1 = fq_mul_ol(a.c1, b.c1);
  Fq t2 = fq_mul_ol(fq_add(a.c0, a.c1), fq_add(b.c0, b.c1));
  return Fq2{fq_sub(t0, t1), fq_sub(fq_sub(t2, t0), t1)};
}

SNARKV_TW Fq2 fq2_sqr(const Fq2& a) {
  Fq t0 = fq_mul_ol(fq_add(a.c0, a.c1), fq_sub(a.c0, a.c1));
  Fq t1 = fq_mul_ol(a.c0, a.c1);
  return Fq2{t0, fq_dbl(t1)};
}

SNARKV_HD Fq2 fq2_mul_fq(const Fq2& a, const Fq& s) { return Fq2{fq_mul_ol(a.c0, s), fq_mul_ol(a.c1, s)}; }

// (a + b u)(9 + u) = (9a - b) + (a + 9b) u
SNARKV_HD Fq2 fq2_mul_xi(const Fq2& a) {
  Fq a8 = fq_dbl(fq_dbl(fq_dbl(a.c0)));
  Fq b8 = fq_dbl(fq_dbl(fq_dbl(a.c1)));
  return Fq2{fq_sub(fq_add(a8, a.c0), a.c1), fq_add(fq_add(b8, a.c1), a.c0)};
}

SNARKV_TW Fq2 fq2_inv(const Fq2& a) {
  Fq d = fq_inv(fq_add(fq_mul_ol(a.c0, a.c0), fq_mul_ol(a.c1, a.c1)));
  return Fq2{fq_mul_ol(a.c0, d), fq_neg(fq_mul_ol(a.c1, d))};
}

// ---------------------------------------------------------------- Fq6
SNARKV_HD Fq6 fq6_zero() { return Fq6{fq2_zero(), fq2_zero(), fq2_zero()}; }
SNARKV_HD Fq6 fq6_one() { return Fq6{fq2_one(), fq2_zero(), fq2_zero()}; }
SNARKV_HD Fq6 fq6_add(const Fq6& a, const Fq6& b) {
  return Fq6{fq2_add(a.c0, b.c0), fq2_add(a.c1, b.c1), fq2_add(a.c2, b.c2)};
}
SNARKV_HD Fq6 fq6_sub(const Fq6& a, const Fq6& b) {
  return Fq6{fq2_sub(a.c0, b.c0), fq2_sub(a.c1, b.c1), fq2_sub(a.c2, b.c2)};
}
SNARKV_HD Fq6 fq6_neg(const Fq6& a) { return Fq6{fq2_neg(a.c0), fq2_neg(a.c1), fq2_neg(a.c2)}; }
SNARKV_HD Fq6 fq6_dbl(const Fq6& a) { return Fq6{fq2_dbl(a.c0), fq2_dbl(a.c1), fq2_dbl(a.c2)}; }
SNARKV_HD bool fq6_eq(const Fq6& a, const Fq6& b) {
  return fq2_eq(a.c0, b.c0) && fq2_eq(a.c1, b.c1) && fq2_eq(a.c2, b.c2);
}
// multiply by v:  (c0, c1, c2) -> (xi c2, c0, c1)
SNARKV_HD Fq6 fq6_mul_v(const Fq6& a) { return Fq6{fq2_mul_xi(a.c2), a.c0, a.c1}; }

SNARKV_TW Fq6 fq6_mul(const Fq6& a, const Fq6& b) {
  Fq2 v0 = fq2_mul(a.c0, b.c0);
  Fq2 v1 = fq2_mul(a.c1, b.c1);
  Fq2 v2 = fq2_mul(a.c2, b.c2);
  Fq2 t0 = fq2_sub(fq2_sub(fq2_mul(fq2_add(a.c1, a.c2), fq2_add(b.c1, b.c2)), v1), v2);
  Fq2 t1 = fq2_sub(fq2_sub(fq2_mul(fq2_add(a.c0, a.c1), fq2_add(b.c0, b.c1)), v0), v1);
  Fq2 t2 = fq2_sub(fq2_sub(fq2_mul(fq2_add(a.c0, a.c2), fq2_add(b.c0, b.c2)), v0), v2);
  return Fq6{fq2_add(v0, fq2_mul_xi(t0)), fq2_add(t1, fq2_mul_xi(v2)), fq2_add(t2, v1)};
}

// a * (b0 + b1 v): five Fq2 products
SNARKV_TW Fq6 fq6_mul_by_01(const Fq6& a, const Fq2& b0, const Fq2& b1) {
  Fq2 v0 = fq2_mul(a.c0, b0);
  Fq2 v1 = fq2_mul(a.c1, b1);
  Fq2 a2b1 = fq2_mul(a.c2, b1);
  Fq2 a2b0 = fq2_mul(a.c2, b0);
  Fq2 t1 = fq2_sub(fq2_sub(fq2_mul(fq2_add(a.c0, a.c1), fq2_add(b0, b1)), v0), v1);
  return Fq6{fq2_add(v0, fq2_mul_xi(a2b1)), t1, fq2_add(a2b0, v1)};
}

SNARKV_TW Fq6 fq6_mul_fq2(const Fq6& a, const Fq2& s) {
  return Fq6{fq2_mul(a.c0, s), fq2_mul(a.c1, s), fq2_mul(a.c2, s)};
}

SNARKV_TW Fq6 fq6_inv(const Fq6& a) {
  Fq2 t0 = fq2_sub(fq2_sqr(a.c0), fq2_mul_xi(fq2_mul(a.c1, a.c2)));
  Fq2 t1 = fq2_sub(fq2_mul_xi(fq2_sqr(a.c2)), fq2_mul(a.c0, a.c1));
  Fq2 t2 = fq2_sub(fq2_sqr(a.c1), fq2_mul(a.c0, a.c2));
  Fq2 d = fq2_add(fq2_mul(a.c0, t0), fq2_mul_xi(fq2_add(fq2_mul(a.c2, t1), fq2_mul(a.c1, t2))));
  Fq2 di = fq2_inv(d);
  return Fq6{fq2_mul(t0, di), fq2_mul(t1, di), fq2_mul(t2, di)};
}

// ---------------------------------------------------------------- Fq12
SNARKV_HD Fq12 fq12_one() { return Fq12{fq6_one(), fq6_zero()}; }
SNARKV_HD bool fq12_eq(const Fq12& a, const Fq12& b) { return fq6_eq(a.c0, b.c0) && fq6_eq(a.c1, b.c1); }
SNARKV_HD bool fq12_is_one(const Fq12& a) { return fq12_eq(a, fq12_one()); }
SNARKV_HD Fq12 fq12_conj(const Fq12& a) { return Fq12{a.c0, fq6_neg(a.c1)}; }

SNARKV_TW Fq12 fq12_mul(const Fq12& a, const Fq12& b) {
  Fq6 v0 = fq6_mul(a.c0, b.c0);
  Fq6 v1 = fq6_mul(a.c1, b.c1);
  Fq6 t = fq6_sub(fq6_sub(fq6_mul(fq6_add(a.c0, a.c1), fq6_add(b.c0, b.c1)), v0), v1);
  return Fq12{fq6_add(v0, fq6_mul_v(v1)), t};
}

// complex squaring: two Fq6 products
SNARKV_TW Fq12 fq12_sqr(const Fq12& a) {
  Fq6 v0 = fq6_mul(a.c0, a.c1);
  Fq6 t = fq6_mul(fq6_add(a.c0, a.c1), fq6_add(a.c0, fq6_mul_v(a.c1)));
  Fq6 c0 = fq6_sub(fq6_sub(t, v0), fq6_mul_v(v0));
  return Fq12{c0, fq6_dbl(v0)};
}

SNARKV_TW Fq12 fq12_inv(const Fq12& a) {
  Fq6 d = fq6_sub(fq6_mul(a.c0, a.c0), fq6_mul_v(fq6_mul(a.c1, a.c1)));
  Fq6 di = fq6_inv(d);
  return Fq12{fq6_mul(a.c0, di), fq6_neg(fq6_mul(a.c1, di))};
}

// f * (l0 + l1 w + l2 w^3), i.e. the sparse element with c0 = (l0,0,0),
// c1 = (l1,l2,0): 13 Fq2 products instead of 18.
SNARKV_TW Fq12 fq12_mul_by_line(const Fq12& f, const Fq2& l0, const Fq2& l1, const Fq2& l2) {
  Fq6 v0 = fq6_mul_fq2(f.c0, l0);
  Fq6 v1 = fq6_mul_by_01(f.c1, l1, l2);
  Fq6 t = fq6_mul_by_01(fq6_add(f.c0, f.c1), fq2_add(l0, l1), l2);
  return Fq12{fq6_add(v0, fq6_mul_v(v1)), fq6_sub(fq6_sub(t, v0), v1)};
}

// p^k-power Frobenius, k in {1,2,3}.  With f = sum_i g_i w^i (g_i in Fq2;
// w^0..w^5 <-> c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2):
// pi^k(f) = sum_i conj^k(g_i) * gamma_{k,i} w^i,  gamma_{k,i} = xi^(i (p^k-1)/6).
struct FrobTable {
  uint32_t g[3][5][2][8];
};
SNARKV_HD Fq2 frob_gamma(int k, int i) {
  constexpr uint32_t g1[5][2][8] = BN254_FROB_GAMMA_1;
  constexpr uint32_t g2[5][2][8] = BN254_FROB_GAMMA_2;
  constexpr uint32_t g3[5][2][8] = BN254_FROB_GAMMA_3;
  Fq2 r;
  for (int l = 0; l < 8; ++l) {
    r.c0.v[l] = (k == 1) ? g1[i - 1][0][l] : (k == 2) ? g2[i - 1][0][l] : g3[i - 1][0][l];
    r.c1.v[l] = (k == 1) ? g1[i - 1][1][l] : (k == 2) ? g2[i - 1][1][l] : g3[i - 1][1][l];
  }
  return r;
}

SNARKV_TW Fq12 fq12_frobenius(const Fq12& f, int k) {
  bool cj = (k & 1) != 0;
  Fq2 g0 = cj ? fq2_conj(f.c0.c0) : f.c0.c0;
  Fq2 g1 = cj ? fq2_conj(f.c1.c0) : f.c1.c0;
  Fq2 g2 = cj ? fq2_conj(f.c0.c1) : f.c0.c1;
  Fq2 g3 = cj ? fq2_conj(f.c1.c1) : f.c1.c1;
  Fq2 g4 = cj ? fq2_conj(f.c0.c2) : f.c0.c2;
  Fq2 g5 = cj ? fq2_conj(f.c1.c2) : f.c1.c2;
  Fq12 r;
  r.c0.c0 = g0;
  r.c1.c0 = fq2_mul(g1, frob_gamma(k, 1));
  r.c0.c1 = fq2_mul(g2, frob_gamma(k, 2));
  r.c1.c1 = fq2_mul(g3, frob_gamma(k, 3));
  r.c0.c2 = fq2_mul(g4, frob_gamma(k, 4));
  r.c1.c2 = fq2_mul(g5, frob_gamma(k, 5));
  return r;
}

}  // namespace snarkv
