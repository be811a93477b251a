// Lane-level pieces of the workgroup-cooperative pairing on the lazy 9x29-bit
// field: the arithmetic the shipped `k_decide` uses.
//
// Why 29-bit here too: on the exact 8x32 field every signed sum of a reduction
// is a dependent carry chain, and gfx950 needs two wait states between a VALU
// writing a carry and the VALU consuming it -- the cheap-looking additions cost
// as much as the products.  In the lazy form a sum is nine independent
// `v_add_u32`, and one float-estimated quotient (`fq29_reduce_small`) squeezes
// the up-to-22-term coefficient back below 1.5p.
//
// Invariant of a coefficient c held in LDS: carry-normalised, |c| < 1.5p, so both
// operands of a product stay inside the multiplier budget (|limb| < 2^29) and every
// fused two-product value inside (-p/16, 17p/16).
#pragma once
#include "fq29.h"

namespace snarkv {

struct Fq2_29 {
  Fq29 c0, c1;
};

SNARKV_HD Fq2_29 frob29_gamma(int k, int i) {
  constexpr int32_t g1[5][2][9] = BN254_FROB29_GAMMA_1;
  constexpr int32_t g2[5][2][9] = BN254_FROB29_GAMMA_2;
  constexpr int32_t g3[5][2][9] = BN254_FROB29_GAMMA_3;
  Fq2_29 r;
  for (int l = 0; l < 9; ++l) {
    r.c0.v[l] = (k == 1) ? g1[i - 1][0][l] : (k == 2) ? g2[i - 1][0][l] : g3[i - 1][0][l];
    r.c1.v[l] = (k == 1) ? g1[i - 1][1][l] : (k == 2) ? g2[i - 1][1][l] : g3[i - 1][1][l];
  }
  return r;
}

// ---------------------------------------------------------------------------
// The round `k_decide` runs ("coop3"; the first version of this round, 204 single
// products + two LDS reduction stages driven by generated tables, was 1.5x slower).
//
// A = sum_i (a[2i] + a[2i+1] u) w^i, same for B.  With V(k') = sum_{i1+i2=k'}
// a_{i1} b_{i2} (an Fq2 value, k' = 0..10) the product is
//     out(k) = V(k) + xi V(k+6)   (k = 0..4),   out(5) = V(5),   xi = 9 + u.
// Lane l = 16 k + 8 e + j  (k = output power of w, e = power of u, j < 6):
//     j <= k : the LOW pair  (i1, i2) = (j, k - j)
//     j >  k : the HIGH pair (i1, i2) = (j, k + 6 - j)       [i1 + i2 = k + 6]
// and each lane computes ONE Fq2-coefficient of its pair as a fused two-product
// Montgomery step (fq29_mul2):   e = 0: a0 b0 - a1 b1     e = 1: a0 b1 + a1 b0.
// 72 active lanes (of 96 = 1.5 wavefronts) instead of 204 products, and the
// reduction needs no LDS: low and high sums are butterflies inside the 8-lane
// group (DPP), the partner coefficient of xi V sits 8 lanes away (row_ror:8),
// and lane j = 0 of each group applies   low + 9 hi_e -/+ hi_(1-e)   in one
// 64-bit carry chain followed by the float-estimated squeeze.
// Invariant of a stored coefficient: carry-normalised, |c| < 1.5p.
struct Coop3Lane {
  int k, e, i1, i2;
  bool active, high;
};

SNARKV_HD Coop3Lane coop3_lane(int lane) {
  Coop3Lane L;
  L.k = lane >> 4;
  L.e = (lane >> 3) & 1;
  int j = lane & 7;
  L.active = L.k < 6 && j < 6;
  L.high = j > L.k;
  L.i1 = j;
  L.i2 = L.high ? L.k + 6 - j : L.k - j;
  if (!L.active) {
    L.i1 = 0;
    L.i2 = 0;
  }
  return L;
}

// the lane's fused product  a0*y0 + (+-a1)*y1  with  y0 = b[2 i2 + e],
// y1 = b[2 i2 + 1 - e] (the caller picks them by address, so both u-powers run
// the same instruction stream):   e = 0: a0 b0 - a1 b1     e = 1: a0 b1 + a1 b0
SNARKV_HD Fq29 coop3_product(int e, const Fq29& a0, const Fq29& a1, const Fq29& y0, const Fq29& y1) {
  Fq29 x1;
  int32_t m = e - 1;  // e = 0: all ones -> negate
#pragma unroll
  for (int i = 0; i < 9; ++i) x1.v[i] = (a1.v[i] ^ m) - m;
  return fq29_mul2(a0, y0, x1, y1);
}

// lo, hi, hp: limb-wise sums of <= 6 carry-normalised products (limbs 0..7
// read as UNSIGNED: up to 6 * 2^29; limb 8 signed); hp = the high sum of the
// other u-power.  Returns low + 9 hi - hp (e = 0) or low + 9 hi + hp (e = 1),
// carry-normalised and squeezed below 1.5p.
SNARKV_HD Fq29 coop3_finalize(int e, const Fq29& lo, const Fq29& hi, const Fq29& hp) {
  Fq29 t;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int64_t x = (int64_t)(uint32_t)lo.v[i] + 9 * (int64_t)(uint32_t)hi.v[i];
    int64_t y = (int64_t)(uint32_t)hp.v[i];
    x += e ? y : -y;
    x += c;
    t.v[i] = (int32_t)x & kMask29;
    c = x >> 29;
  }
  {
    int64_t x = (int64_t)lo.v[8] + 9 * (int64_t)hi.v[8] + (e ? (int64_t)hp.v[8] : -(int64_t)hp.v[8]) + c;
    t.v[8] = (int32_t)x;
  }
  return fq29_reduce_small(t);
}

}  // namespace snarkv
