// BN254 Fq for the MSM hot loops: 9 x 29-bit SIGNED limbs, lazy carries,
// Montgomery with R = 2^261.
//
// Why not the 8 x 32 form of fq.cuh: measured on MI355X (tools/ubench_fq.hip,
// profiles/r01_ubench.txt) `v_mad_u64_u32` issues at the SAME rate as any other
// VOP3 instruction (~4.5 cycles per wave-instruction at >= 2 waves/SIMD), so a
// saturated 32-bit limb pays as much for carry handling (a 64-bit add plus
// zero-extension moves per product: 128 mad + 128 v_lshl_add_u64 + ~370 v_mov
// per product in the compiled 8x32 CIOS) as for the multiplies.  With 29-bit
// limbs a 64-bit column accumulator absorbs all 18 partial products of a column
// without any carry: ONE `v_mad_i64_i32` per partial product, ~205 instructions
// per Montgomery product instead of ~660.  Signed limbs make a - b limb-wise
// (9 VOP2 subtracts, no borrow chain, no +kp offset).
//
// Contract ("lazy" values):
//   * value = sum l_i 2^(29 i), any integer in (-8p, 8p) congruent to the field
//     element; limb 8 carries the sign/overflow.
//   * mul(a, b) needs  9 * max|a_i| * max|b_j| < 2^63 - 2^61.2, i.e.
//     max|a_i| * max|b_j| < 2^59.6 (one operand carry-normalised (< 2^29), the
//     other up to 2^30.6; or both < 2^29.8).  Its result has limbs 0..7 in
//     [0, 2^29) and value in (-p/8, p + p/8) for operands within (-8p, 8p)... see
//     the bound notes at each call site in g1_29.cuh.
//   * add/sub/neg are limb-wise and never carry; `fq29_norm` re-normalises the
//     limbs (value unchanged) when the next product needs it.
//   * zero/equality tests mod p need `fq29_is_zero_mod_p` (canonicalising).
// Memory form: 9 x int32 (36 bytes).  Same source compiles for the host
// (tests/hosttest).
#pragma once
#include <stdint.h>
#include "fq.cuh"

namespace snarkv {

struct Fq29 {
  int32_t v[9];
};

constexpr int32_t kMask29 = (1 << 29) - 1;

// radix-2^29 constants (BN254_P29_LIMBS, BN254_P29_NINV, BN254_ONE29_LIMBS,
// BN254_R2_29_LIMBS) come from bn254_consts.h (gen_consts.py).

SNARKV_HD int32_t fq29_p(int i) {
  constexpr int32_t p[9] = BN254_P29_LIMBS;
  return p[i];
}

SNARKV_HD Fq29 fq29_zero() {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = 0;
  return r;
}

SNARKV_HD Fq29 fq29_one() {
  constexpr int32_t c[9] = BN254_ONE29_LIMBS;
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = c[i];
  return r;
}

// exact all-limbs-zero test (the stored identity marker), NOT a mod-p test
SNARKV_HD bool fq29_limbs_all_zero(const Fq29& a) {
  int32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) acc |= a.v[i];
  return acc == 0;
}

SNARKV_HD Fq29 fq29_add(const Fq29& a, const Fq29& b) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}

SNARKV_HD Fq29 fq29_sub(const Fq29& a, const Fq29& b) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] - b.v[i];
  return r;
}

SNARKV_HD Fq29 fq29_neg(const Fq29& a) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = -a.v[i];
  return r;
}

SNARKV_HD Fq29 fq29_dbl(const Fq29& a) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] * 2;
  return r;
}

// carry-normalise: limbs 0..7 -> [0, 2^29), limb 8 keeps sign; value unchanged
SNARKV_HD Fq29 fq29_norm(const Fq29& a) {
  Fq29 r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int32_t t = a.v[i] + c;
    r.v[i] = t & kMask29;
    c = t >> 29;  // arithmetic
  }
  r.v[8] = a.v[8] + c;
  return r;
}

// Montgomery product a*b*2^-261 (mod p), column-wise (product scanning) with a
// single 64-bit accumulator: 81 + 81 `v_mad_i64_i32`, 17 64-bit shifts, 9
// `v_mul_lo_u32`.
SNARKV_HD Fq29 fq29_mul(const Fq29& a, const Fq29& b) {
  int32_t m[9];
  Fq29 r;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)BN254_P29_NINV) & (uint32_t)kMask29);
    acc += (int64_t)m[k] * fq29_p(0);
    acc >>= 29;  // low 29 bits are zero
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}

// a*b + c*d with ONE Montgomery reduction (the two halves of an Fq2 product
// coefficient).  All four inputs carry-normalised or the limb-wise negation of
// a carry-normalised value (|limb| < 2^29): a column then holds at most
// 18 + 9 products of magnitude < 2^58, below 2^63.  243 mads instead of 326.
SNARKV_HD Fq29 fq29_mul2(const Fq29& a, const Fq29& b, const Fq29& c, const Fq29& d) {
  int32_t m[9];
  Fq29 r;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)c.v[i] * d.v[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)BN254_P29_NINV) & (uint32_t)kMask29);
    acc += (int64_t)m[k] * fq29_p(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)c.v[i] * d.v[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}

// a^2; a must be carry-normalised (|limb| < 2^29): doubled limbs stay < 2^30.
SNARKV_HD Fq29 fq29_sqr(const Fq29& a) {
  int32_t m[9], a2[9];
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) a2[i] = a.v[i] * 2;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc += (int64_t)a2[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (int64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)BN254_P29_NINV) & (uint32_t)kMask29);
    acc += (int64_t)m[k] * fq29_p(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; 2 * i < k; ++i) acc += (int64_t)a2[i] * a.v[k - i];
    if ((k & 1) == 0) acc += (int64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}

// Unique representative in [0, p), carry-normalised.  x must be within
// (-8p, 8p); one Montgomery product by 2^261 (i.e. by `one`) squeezes the
// value into (-p/8, 9p/8) without changing the residue, then at most one +p and
// one -p.
SNARKV_HD Fq29 fq29_canon_residue(const Fq29& x) {
  Fq29 y = fq29_mul(fq29_norm(x), fq29_one());  // same residue: x * R * R^-1
  // y limbs 0..7 in [0,2^29); y.v[8] small signed
  Fq29 t;
  int32_t neg = y.v[8] >> 31;  // all ones if negative
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = y.v[i] + (fq29_p(i) & neg);
  t = fq29_norm(t);
  // subtract p if t >= p
  Fq29 d;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int32_t s = t.v[i] - fq29_p(i) + c;
    d.v[i] = s & kMask29;
    c = s >> 29;
  }
  d.v[8] = t.v[8] - fq29_p(8) + c;
  int32_t keep = d.v[8] >> 31;  // negative -> t < p -> keep t
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = (t.v[i] & keep) | (d.v[i] & ~keep);
  return t;
}

SNARKV_HD bool fq29_is_zero_mod_p(const Fq29& x) { return fq29_limbs_all_zero(fq29_canon_residue(x)); }

// boundary codecs: 8 x u32 canonical integer <-> Montgomery (R = 2^261) limbs
SNARKV_HD Fq29 fq29_from_canonical(const uint32_t w[8]) {
  Fq29 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 29 * i;
    int word = bit >> 5, sh = bit & 31;
    uint64_t v = w[word];
    if (word + 1 < 8) v |= (uint64_t)w[word + 1] << 32;
    a.v[i] = (int32_t)((uint32_t)(v >> sh) & (uint32_t)kMask29);
  }
  constexpr int32_t r2[9] = BN254_R2_29_LIMBS;
  Fq29 b;
#pragma unroll
  for (int i = 0; i < 9; ++i) b.v[i] = r2[i];
  return fq29_mul(a, b);
}

SNARKV_HD void fq29_to_canonical(const Fq29& a, uint32_t w[8]) {
  Fq29 one_raw = fq29_zero();
  one_raw.v[0] = 1;
  Fq29 y = fq29_mul(fq29_norm(a), one_raw);  // a * R^-1: out of Montgomery form
  // canonicalise the plain integer residue
  Fq29 t;
  int32_t neg = y.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = y.v[i] + (fq29_p(i) & neg);
  t = fq29_norm(t);
  Fq29 d;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int32_t s = t.v[i] - fq29_p(i) + c;
    d.v[i] = s & kMask29;
    c = s >> 29;
  }
  d.v[8] = t.v[8] - fq29_p(8) + c;
  int32_t keep = d.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = (t.v[i] & keep) | (d.v[i] & ~keep);
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 29 * i;
    int word = bit >> 5, sh = bit & 31;
    uint64_t v = (uint64_t)(uint32_t)t.v[i] << sh;
    w[word] |= (uint32_t)v;
    if (word + 1 < 8) w[word + 1] |= (uint32_t)(v >> 32);
  }
}

// x - round(x/p) p for |x| up to ~64p, x carry-normalised: the quotient is
// estimated from the top limb (bits 232..), exact to +-1, so |result| < 1.5p.
// Result carry-normalised.  (Cheap modular squeeze after a lazy sum of many
// products; one float multiply, nine 64-bit mads.)
SNARKV_HD Fq29 fq29_reduce_small(const Fq29& x) {
  const float inv_ptop = 1.0f / (float)(0x0030644e);  // p >> 232
  float qf = (float)x.v[8] * inv_ptop;
  int32_t q = (int32_t)(qf + (qf >= 0 ? 0.5f : -0.5f));
  Fq29 r;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int64_t t = (int64_t)x.v[i] - (int64_t)q * fq29_p(i) + c;
    r.v[i] = (int32_t)t & kMask29;
    c = t >> 29;
  }
  r.v[8] = (int32_t)((int64_t)x.v[8] - (int64_t)q * fq29_p(8) + c);
  return r;
}

// carry-normalised k*x for a small constant k (|k| < 2^20), x carry-normalised
SNARKV_HD Fq29 fq29_mul_small_norm(const Fq29& x, int32_t k) {
  Fq29 r;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int64_t t = (int64_t)x.v[i] * k + c;
    r.v[i] = (int32_t)t & kMask29;
    c = t >> 29;
  }
  r.v[8] = (int32_t)((int64_t)x.v[8] * k + c);
  return r;
}

// a^(p-2) (lane-uniform exponent); a must be carry-normalised, result too.
// Kept as the independent cross-check of fq29_inv (tests/hosttest).
SNARKV_HD_NOINLINE Fq29 fq29_inv_fermat(const Fq29& a) {
  constexpr uint32_t e[8] = BN254_P_MINUS_2_LIMBS;
  Fq29 res = fq29_one();
  for (int i = 7; i >= 0; --i) {
    uint32_t w = e[i];
    for (int b = 31; b >= 0; --b) {
      res = fq29_sqr(fq29_norm(res));
      if ((w >> b) & 1u) res = fq29_mul(res, a);
    }
  }
  return res;
}

// ---- binary extended Euclid on plain 256-bit integers (8 x u32, little-endian) ----
// Every inversion on this path sits on a one-lane latency chain (to_affine at
// the end of an MSM, the norm at the bottom of the Fq12 inversion), where the
// 380 dependent field products of Fermat cost ~0.19 ms; ~750 shift/subtract
// steps of ~30 integer instructions are ~3.5x shorter.
struct U256w {
  uint32_t w[8];
};

SNARKV_HD void u256_shr1(U256w& a) {
#pragma unroll
  for (int i = 0; i < 7; ++i) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 31);
  a.w[7] >>= 1;
}

// a -= b, returns the borrow (1 if a < b)
SNARKV_HD uint32_t u256_sub(U256w& a, const U256w& b) {
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a.w[i] - b.w[i] - br;
    a.w[i] = (uint32_t)t;
    br = (t >> 32) & 1u;
  }
  return (uint32_t)br;
}

// a += (p & mask)
SNARKV_HD void u256_add_p_masked(U256w& a, uint32_t mask) {
  constexpr uint32_t pl[8] = BN254_P_LIMBS;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t t = (uint64_t)a.w[i] + (pl[i] & mask) + c;
    a.w[i] = (uint32_t)t;
    c = t >> 32;
  }
}

SNARKV_HD bool u256_is_one(const U256w& a) {
  uint32_t r = a.w[0] ^ 1u;
#pragma unroll
  for (int i = 1; i < 8; ++i) r |= a.w[i];
  return r == 0;
}

// x/2 mod p for x in [0, p)   (x + p < 2^255: no carry out)
SNARKV_HD void u256_half_mod_p(U256w& x) {
  u256_add_p_masked(x, 0u - (x.w[0] & 1u));
  u256_shr1(x);
}

// x = x - y mod p for x, y in [0, p)
SNARKV_HD void u256_sub_mod_p(U256w& x, const U256w& y) {
  uint32_t br = u256_sub(x, y);
  u256_add_p_masked(x, 0u - br);
}

// a^-1 mod p for a in [1, p); 0 for a = 0.  Plain integers in, plain integer out.
SNARKV_HD_NOINLINE void fq_words_inv_binary(const uint32_t a[8], uint32_t out[8]) {
  constexpr uint32_t pl[8] = BN254_P_LIMBS;
  U256w u, v, x1, x2;
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u.w[i] = a[i];
    v.w[i] = pl[i];
    x1.w[i] = 0;
    x2.w[i] = 0;
    nz |= a[i];
  }
  x1.w[0] = 1;
  if (nz == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = 0;
    return;
  }
  // invariants: u = x1 * a, v = x2 * a (mod p); gcd(u, v) = 1; v stays odd
  while (!u256_is_one(u) && !u256_is_one(v)) {
    if ((u.w[0] & 1u) == 0) {
      u256_shr1(u);
      u256_half_mod_p(x1);
    } else if ((v.w[0] & 1u) == 0) {
      u256_shr1(v);
      u256_half_mod_p(x2);
    } else {
      U256w d = u;
      uint32_t br = u256_sub(d, v);
      if (br == 0) {  // u >= v
        u = d;
        u256_sub_mod_p(x1, x2);
      } else {
        u256_sub(v, u);
        u256_sub_mod_p(x2, x1);
      }
    }
  }
  bool uo = u256_is_one(u);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = uo ? x1.w[i] : x2.w[i];
}

// Field inverse in the Montgomery domain; a within (-8p, 8p), result
// carry-normalised.  0 -> 0 (as Fermat's a^(p-2) gives).
SNARKV_HD Fq29 fq29_inv(const Fq29& a) {
  uint32_t w[8], r[8];
  fq29_to_canonical(a, w);
  fq_words_inv_binary(w, r);
  return fq29_from_canonical(r);
}

}  // namespace snarkv
