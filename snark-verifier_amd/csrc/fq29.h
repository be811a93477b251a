// BN254 Fq for the MSM hot loops: 9 x 29-bit SIGNED limbs, lazy carries,
// Montgomery with R = 2^261.
//
// Why not the 8 x 32 form of fq.h: measured on MI355X (tools/ubench_fq.hip,
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
//     the bound notes at each call site in g1_29.h.
//   * add/sub/neg are limb-wise and never carry; `fq29_norm` re-normalises the
//     limbs (value unchanged) when the next product needs it.
//   * zero/equality tests mod p need `fq29_is_zero_mod_p` (canonicalising).
// Memory form: 9 x int32 (36 bytes).  Same source compiles for the host
// (tests/hosttest).
#pragma once
#include <stdint.h>
#include "fq.h"

namespace snarkv {

struct Fq29 {
  int32_t v[9];
};

constexpr int32_t kMask29 = (1 << 29) - 1;

// radix-2^29 constants (SNARKV_FQ29_P_LIMBS, SNARKV_FQ29_NINV, SNARKV_FQ29_ONE_LIMBS,
// SNARKV_FQ29_R2_LIMBS) come from bn254_consts.h (gen_consts.py).

SNARKV_HD constexpr int32_t fq29_p(int i) {
  constexpr int32_t p[9] = SNARKV_FQ29_P_LIMBS;
  return p[i];
}

SNARKV_HD Fq29 fq29_zero() {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = 0;
  return r;
}

SNARKV_HD Fq29 fq29_one() {
  constexpr int32_t c[9] = SNARKV_FQ29_ONE_LIMBS;
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

// How the products are issued on the device.  The compiler, knowing that masked limbs are non-negative,
// turns a mixed-sign 32x32+64 product into `v_mad_u64_u32` plus a sign fix-up (shift, move, subtract) and
// keeps the operand products and the reduction products in two chains joined by a 64-bit add per column:
// 2 642 instructions per mixed G1 addition where 1 475 are multiply-adds -- and on this machine every VOP3
// instruction costs the same issue slot (profiles/r01_ubench_isa_rates.txt).  So fq29_mul / fq29_mul2 /
// fq29_sqr take their device bodies from gen_fq29_mul_asm.py: every column of the product scanning loop is
// ONE chain of `v_mad_i64_i32` on the column accumulator (carry-out to VCC, dead), 2 276 instructions per
// entry of k_accumulate.  Stepping stones (rounds 1-2, measured): one asm statement per operand product
// (-12 %), one statement per product of either kind (slower again: the compiler pads consecutive
// VCC-writing asm statements with s_nop), one statement per column (this form).
// -DSNARKV_NO_SMAD_ASM keeps the plain C below (host builds always use it).  The pairing (decider.hip), a few
// latency-bound wavefronts, lost 3 % to the one-statement-per-product form but gains 2-3 % from the column
// chains (decide 0.887 -> 0.866 ms, 1 024 accumulators 1.51 -> 1.46 ms); Poseidon's Fr products stay plain C.
SNARKV_HD int64_t fq29_smad(int32_t a, int32_t b, int64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SNARKV_NO_SMAD_ASM)
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  return acc;
#else
  return acc + (int64_t)a * b;
#endif
}

// Montgomery product a*b*2^-261 (mod p), column-wise (product scanning) with a
// single 64-bit accumulator: 81 + 81 `v_mad_i64_i32`, 17 64-bit shifts, 9
// `v_mul_lo_u32`, 18 masks.
SNARKV_HD Fq29 fq29_mul(const Fq29& a, const Fq29& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SNARKV_NO_SMAD_ASM)
#include "fq29_mul_asm.inc"  // every column one chain of v_mad_i64_i32 (gen_fq29_mul_asm.py)
#else
  int32_t m[9];
  Fq29 r;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc = fq29_smad(a.v[i], b.v[k - i], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)SNARKV_FQ29_NINV) & (uint32_t)kMask29);
    acc += (int64_t)m[k] * fq29_p(0);
    acc >>= 29;  // low 29 bits are zero
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc = fq29_smad(a.v[i], b.v[k - i], acc);
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
#endif
}

// a*b + c*d with ONE Montgomery reduction (the two halves of an Fq2 product
// coefficient).  All four inputs carry-normalised or the limb-wise negation of
// a carry-normalised value (|limb| < 2^29): a column then holds at most
// 18 + 9 products of magnitude < 2^58, below 2^63.  243 mads instead of 326.
SNARKV_HD Fq29 fq29_mul2(const Fq29& a, const Fq29& b, const Fq29& c, const Fq29& d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SNARKV_NO_SMAD_ASM)
#include "fq29_mul2_asm.inc"
#else
  int32_t m[9];
  Fq29 r;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc = fq29_smad(a.v[i], b.v[k - i], acc);
#pragma unroll
    for (int i = 0; i <= k; ++i) acc = fq29_smad(c.v[i], d.v[k - i], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)SNARKV_FQ29_NINV) & (uint32_t)kMask29);
    acc += (int64_t)m[k] * fq29_p(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc = fq29_smad(a.v[i], b.v[k - i], acc);
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc = fq29_smad(c.v[i], d.v[k - i], acc);
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
#endif
}

// a^2; a must be carry-normalised (|limb| < 2^29): doubled limbs stay < 2^30.
SNARKV_HD Fq29 fq29_sqr(const Fq29& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SNARKV_NO_SMAD_ASM)
#include "fq29_sqr_asm.inc"
#else
  int32_t m[9], a2[9];
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) a2[i] = a.v[i] * 2;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc = fq29_smad(a2[i], a.v[k - i], acc);
    if ((k & 1) == 0) acc = fq29_smad(a.v[k / 2], a.v[k / 2], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)SNARKV_FQ29_NINV) & (uint32_t)kMask29);
    acc += (int64_t)m[k] * fq29_p(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; 2 * i < k; ++i) acc = fq29_smad(a2[i], a.v[k - i], acc);
    if ((k & 1) == 0) acc = fq29_smad(a.v[k / 2], a.v[k / 2], acc);
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)m[i] * fq29_p(k - i);
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
#endif
}

// (Measured and removed, round 2: PAIRS of independent products interleaved column by column -- 2 404 -> 2 278 issued
// instructions per bucket addition, one MSM alone level, four in flight 1-5 % slower: profiles/r02_ab_pairs.txt, git tag
// exp/madd-pairs.)

// Unique representative in [0, p), carry-normalised.
// canonical limbs of a value y in (-p, 2p) whose limbs 0..7 are already in [0, 2^29) -- in particular any OUTPUT of
// fq29_mul / fq29_mul2 / fq29_sqr (value in (-p/8, p + p/8)): one conditional +p, one conditional -p, no product
SNARKV_HD Fq29 fq29_canon_of_product(const Fq29& y) {
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

// canonical limbs of ANY lazy value (the contract's (-8p, 8p)): one Montgomery product by 2^261 (i.e. by `one`) squeezes
// the value into (-p/8, 9p/8) without changing the residue (x * R * R^-1), then at most one +p and one -p
SNARKV_HD Fq29 fq29_canon_residue(const Fq29& x) { return fq29_canon_of_product(fq29_mul(fq29_norm(x), fq29_one())); }

SNARKV_HD bool fq29_is_zero_mod_p(const Fq29& x) { return fq29_limbs_all_zero(fq29_canon_residue(x)); }

// boundary codecs: 8 x u32 words <-> Montgomery (R = 2^261) limbs.  The words are the canonical integer a < p
// (`PrimeField::to_repr`, the wire form) or, with `mont`, halo2curves' IN-MEMORY form: the four u64 limbs of
// a * 2^256 mod p.  Either way ONE product by a constant: 2^522 / 2^266 on the way in, 1 / 2^256 on the way out.
SNARKV_HD Fq29 fq29_limbs_of_words(const uint32_t w[8]) {
  Fq29 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 29 * i;
    int word = bit >> 5, sh = bit & 31;
    uint64_t v = w[word];
    if (word + 1 < 8) v |= (uint64_t)w[word + 1] << 32;
    a.v[i] = (int32_t)((uint32_t)(v >> sh) & (uint32_t)kMask29);
  }
  return a;
}
SNARKV_HD Fq29 fq29_from_words(const uint32_t w[8], bool mont) {
  constexpr int32_t r2[9] = SNARKV_FQ29_R2_LIMBS;
  constexpr int32_t m_in[9] = SNARKV_FQ29_M256_IN_LIMBS;
  Fq29 b;
#pragma unroll
  for (int i = 0; i < 9; ++i) b.v[i] = mont ? m_in[i] : r2[i];
  return fq29_mul(fq29_limbs_of_words(w), b);
}
SNARKV_HD Fq29 fq29_from_canonical(const uint32_t w[8]) { return fq29_from_words(w, false); }

SNARKV_HD void fq29_to_words(const Fq29& a, uint32_t w[8], bool mont) {
  constexpr int32_t m_out[9] = SNARKV_FQ29_M256_OUT_LIMBS;
  Fq29 mult;
#pragma unroll
  for (int i = 0; i < 9; ++i) mult.v[i] = mont ? m_out[i] : (i == 0 ? 1 : 0);
  Fq29 y = fq29_mul(fq29_norm(a), mult);  // a * R^-1 [* 2^256]: out of the 2^261 domain
  // canonicalise the residue
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
SNARKV_HD void fq29_to_canonical(const Fq29& a, uint32_t w[8]) { fq29_to_words(a, w, false); }

// x - round(x/p) p for |x| up to ~64p, x carry-normalised: the quotient is
// estimated from the top limb (bits 232..), exact to +-1, so |result| < 1.5p.
// Result carry-normalised.  (Cheap modular squeeze after a lazy sum of many
// products; one float multiply, nine 64-bit mads.)
SNARKV_HD Fq29 fq29_reduce_small(const Fq29& x) {
  const float inv_ptop = 1.0f / (float)fq29_p(8);  // p >> 232
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
  constexpr uint32_t e[8] = SNARKV_FQ_P_MINUS_2_LIMBS;
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
  constexpr uint32_t pl[8] = SNARKV_FQ_P_LIMBS;
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
  constexpr uint32_t pl[8] = SNARKV_FQ_P_LIMBS;
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

// ---- safegcd (Bernstein-Yang divsteps), 30 steps per batch -------------------
// The binary Euclid above spends ~750 branchy 256-bit iterations; here the
// branchy part runs on the low 30 bits only (a 2x2 transition matrix per batch)
// and the wide values see two matrix-vector products per batch: ~22 batches of
// ~500 instructions.  Signed 30-bit limbs, 9 per value.
struct S30 {
  int32_t v[9];
};
constexpr int32_t kMask30 = (1 << 30) - 1;

SNARKV_HD S30 s30_modulus() {
  constexpr uint32_t pl[8] = SNARKV_FQ_P_LIMBS;
  S30 m;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 30 * i, word = bit >> 5, sh = bit & 31;
    uint64_t t = word < 8 ? pl[word] : 0;
    if (word + 1 < 8) t |= (uint64_t)pl[word + 1] << 32;
    m.v[i] = (int32_t)((uint32_t)(t >> sh) & (uint32_t)kMask30);
  }
  return m;
}

// p^-1 mod 2^30 (Newton on the low word)
SNARKV_HD uint32_t s30_modulus_inv30() {
  constexpr uint32_t pl[8] = SNARKV_FQ_P_LIMBS;
  uint32_t x = pl[0];  // correct to 3 bits for odd p
#pragma unroll
  for (int i = 0; i < 5; ++i) x *= 2u - pl[0] * x;
  return x & (uint32_t)kMask30;
}

// 30 divsteps on the low bits; t = (u v; q r) with 2^30 (f', g') = t (f, g).  Returns the new delta.
SNARKV_HD int32_t s30_divsteps30(int32_t delta, uint32_t f, uint32_t g, int32_t t[4]) {
  int32_t u = 1, v = 0, q = 0, r = 1;
  for (int i = 0; i < 30; ++i) {
    if (g & 1u) {
      if (delta > 0) {
        delta = -delta;
        uint32_t tf = f;
        f = g;
        g = g - tf;
        int32_t tu = u, tv = v;
        u = q;
        v = r;
        q = q - tu;
        r = r - tv;
      } else {
        g += f;
        q += u;
        r += v;
      }
    }
    delta += 1;
    g >>= 1;
    u *= 2;
    v *= 2;
  }
  t[0] = u;
  t[1] = v;
  t[2] = q;
  t[3] = r;
  return delta;
}

// (f, g) <- t (f, g) / 2^30, exact
SNARKV_HD void s30_update_fg(S30& f, S30& g, const int32_t t[4]) {
  const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
  int64_t cf = u * f.v[0] + v * g.v[0];
  int64_t cg = q * f.v[0] + r * g.v[0];
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cf += u * f.v[i] + v * g.v[i];
    cg += q * f.v[i] + r * g.v[i];
    f.v[i - 1] = (int32_t)cf & kMask30;
    g.v[i - 1] = (int32_t)cg & kMask30;
    cf >>= 30;
    cg >>= 30;
  }
  f.v[8] = (int32_t)cf;
  g.v[8] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^30 mod p; d, e stay in (-2p, p)
SNARKV_HD void s30_update_de(S30& d, S30& e, const int32_t t[4], const S30& m, uint32_t minv30) {
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
  int32_t md = (u & sd) + (v & se);
  int32_t me = (q & sd) + (r & se);
  int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
  int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
  md -= (int32_t)((minv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)kMask30);
  me -= (int32_t)((minv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)kMask30);
  cd += (int64_t)m.v[0] * md;
  ce += (int64_t)m.v[0] * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i] + (int64_t)m.v[i] * md;
    ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i] + (int64_t)m.v[i] * me;
    d.v[i - 1] = (int32_t)cd & kMask30;
    e.v[i - 1] = (int32_t)ce & kMask30;
    cd >>= 30;
    ce >>= 30;
  }
  d.v[8] = (int32_t)cd;
  e.v[8] = (int32_t)ce;
}

// a^-1 mod p for a in [0, p) (0 -> 0); plain integers, 8 x u32 little-endian
// (by value: array parameters of a non-inlined device function live on the stack -- 80 bytes of scratch per lane in every
// kernel that inverts: profiles/r03_kernel_resource_usage.txt)
SNARKV_HD_NOINLINE U256w fq_words_inv_safegcd_v(const U256w av) {
  const uint32_t* a = av.w;
  U256w outv;
  uint32_t* out = outv.w;
  const S30 m = s30_modulus();
  const uint32_t minv30 = s30_modulus_inv30();
  S30 f = m, g, d, e;
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 30 * i, word = bit >> 5, sh = bit & 31;
    uint64_t t = word < 8 ? a[word] : 0;
    if (word + 1 < 8) t |= (uint64_t)a[word + 1] << 32;
    g.v[i] = (int32_t)((uint32_t)(t >> sh) & (uint32_t)kMask30);
    d.v[i] = 0;
    e.v[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) nz |= a[i];
  e.v[0] = 1;
  if (nz != 0) {
    int32_t delta = 1;
    for (int batch = 0; batch < 40; ++batch) {  // <= 25 batches cover the 741-divstep bound for 256 bits
      int32_t t[4];
      delta = s30_divsteps30(delta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
      s30_update_de(d, e, t, m, minv30);
      s30_update_fg(f, g, t);
      int32_t gz = 0;
#pragma unroll
      for (int i = 0; i < 9; ++i) gz |= g.v[i];
      if (gz == 0) break;
    }
  }
  // g = 0, f = +-1 (gcd), d = +-a^-1 in (-2p, p): fix the sign, then into [0, p)
  const int32_t fneg = f.v[8] >> 31;
  int64_t c = 0;
  S30 x;
#pragma unroll
  for (int i = 0; i < 9; ++i) {  // x = fneg ? -d : d
    int64_t tt = (int64_t)((d.v[i] ^ fneg) - fneg) + c;
    x.v[i] = i < 8 ? ((int32_t)tt & kMask30) : (int32_t)tt;
    c = i < 8 ? (tt >> 30) : 0;
  }
  for (int pass = 0; pass < 2; ++pass) {  // x in (-2p, 2p): add p while negative ...
    const int32_t neg = x.v[8] >> 31;
    c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      int64_t tt = (int64_t)x.v[i] + (m.v[i] & neg) + c;
      x.v[i] = i < 8 ? ((int32_t)tt & kMask30) : (int32_t)tt;
      c = i < 8 ? (tt >> 30) : 0;
    }
  }
  {  // ... then subtract p once if x >= p
    S30 y;
    c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      int64_t tt = (int64_t)x.v[i] - m.v[i] + c;
      y.v[i] = i < 8 ? ((int32_t)tt & kMask30) : (int32_t)tt;
      c = i < 8 ? (tt >> 30) : 0;
    }
    const int32_t keep = y.v[8] >> 31;  // negative: x < p
#pragma unroll
    for (int i = 0; i < 9; ++i) x.v[i] = (x.v[i] & keep) | (y.v[i] & ~keep);
  }
  if (nz == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) x.v[i] = 0;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 30 * i, word = bit >> 5, sh = bit & 31;
    uint64_t tt = (uint64_t)(uint32_t)x.v[i] << sh;
    if (word < 8) out[word] |= (uint32_t)tt;
    if (word + 1 < 8) out[word + 1] |= (uint32_t)(tt >> 32);
  }
  return outv;
}
SNARKV_HD void fq_words_inv_safegcd(const uint32_t a[8], uint32_t out[8]) {
  U256w x;
#pragma unroll
  for (int i = 0; i < 8; ++i) x.w[i] = a[i];
  const U256w r = fq_words_inv_safegcd_v(x);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = r.w[i];
}

// Field inverse in the Montgomery domain; a within (-8p, 8p), result
// carry-normalised.  0 -> 0 (as Fermat's a^(p-2) gives).
SNARKV_HD Fq29 fq29_inv(const Fq29& a) {
  uint32_t w[8], r[8];
  fq29_to_canonical(a, w);
  fq_words_inv_safegcd(w, r);
  return fq29_from_canonical(r);
}

}  // namespace snarkv
