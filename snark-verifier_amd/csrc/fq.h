// BN254 base field Fq: 254-bit Montgomery arithmetic on 8 x 32-bit limbs.
//
// This replaces, on the device, what the reference takes from
// `halo2curves::bn256::Fq` (reference `snark-verifier/src/util/arithmetic.rs:13-18`
// re-exports; halo2curves keeps 4 x u64 Montgomery limbs with R = 2^256 -- the
// same residue system, so Montgomery images are bit-identical, only the limb
// width differs).  gfx950 has no 64x64 multiplier: the widest integer multiply
// is `v_mad_u64_u32` (32x32+64 -> 64), so the natural limb is 32 bits.
//
// All functions are plain integer C++ (no intrinsics) so the same source is
// also compiled for the host by tests/ (device-function unit tests without a
// GPU).  Values are always fully reduced to [0, p).
#pragma once
#include <stdint.h>
#include "curve_consts.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SNARKV_HD __host__ __device__ __forceinline__
#define SNARKV_HD_NOINLINE static __host__ __device__ __noinline__
#else
#define SNARKV_HD inline
#define SNARKV_HD_NOINLINE inline
#endif

namespace snarkv {

struct Fq {
  uint32_t v[8];
};

// Inline-constant tables (the compiler folds these into literals / s_mov).
SNARKV_HD constexpr uint32_t fq_p(int i) {
  constexpr uint32_t p[8] = SNARKV_FQ_P_LIMBS;
  return p[i];
}

SNARKV_HD Fq fq_zero() {
  Fq r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}

SNARKV_HD Fq fq_one() {
  constexpr uint32_t c[8] = SNARKV_FQ_ONE_MONT;
  Fq r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c[i];
  return r;
}

SNARKV_HD bool fq_is_zero(const Fq& a) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc |= a.v[i];
  return acc == 0;
}

SNARKV_HD bool fq_eq(const Fq& a, const Fq& b) {
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc |= a.v[i] ^ b.v[i];
  return acc == 0;
}

// r = t - p if t >= p else t   (t < 2p)
SNARKV_HD void fq_reduce_once(uint32_t t[8]) {
  uint32_t d[8];
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)t[i] - fq_p(i) - borrow;
    d[i] = (uint32_t)x;
    borrow = (x >> 32) & 1u;
  }
  // borrow == 1  <=>  t < p  -> keep t
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = borrow ? t[i] : d[i];
}

SNARKV_HD Fq fq_add(const Fq& a, const Fq& b) {
  Fq r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.v[i] + b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  // a + b < 2p < 2^255: no carry out of limb 7
  fq_reduce_once(r.v);
  return r;
}

SNARKV_HD Fq fq_sub(const Fq& a, const Fq& b) {
  Fq r;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)a.v[i] - b.v[i] - borrow;
    r.v[i] = (uint32_t)x;
    borrow = (x >> 32) & 1u;
  }
  // if borrowed add p back
  uint32_t mask = (uint32_t)0 - (uint32_t)borrow;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)r.v[i] + (fq_p(i) & mask);
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return r;
}

SNARKV_HD Fq fq_neg(const Fq& a) {
  Fq z = fq_zero();
  return fq_sub(z, a);
}

SNARKV_HD Fq fq_dbl(const Fq& a) { return fq_add(a, a); }

// Montgomery product a*b*R^-1 mod p -- portable form.  CIOS with the "no-carry"
// merge that is valid because p < 2^255 (top bit of limb 7 clear): the running
// carries A (from a_i*b_j) and C (from m*p_j) never overflow a 32-bit word when
// summed.  This is what the HOST build (tests/hosttest) runs.
SNARKV_HD Fq fq_mul_portable(const Fq& a, const Fq& b) {
  uint32_t t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t A = (uint64_t)a.v[i] * b.v[0] + t[0];
    uint32_t m = (uint32_t)A * SNARKV_FQ_P_INV32;
    uint64_t C = (uint64_t)m * fq_p(0) + (uint32_t)A;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      A = (uint64_t)a.v[i] * b.v[j] + t[j] + (A >> 32);
      C = (uint64_t)m * fq_p(j) + (uint32_t)A + (C >> 32);
      t[j - 1] = (uint32_t)C;
    }
    t[7] = (uint32_t)((C >> 32) + (A >> 32));
  }
  fq_reduce_once(t);
  Fq r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Device form: column-wise (product-scanning) Montgomery with a 96-bit column
// accumulator; one `v_mad_u64_u32` + one `v_addc_co_u32` per partial product,
// carries software-pipelined through SGPR pairs (gen_fq_mul_asm.py explains
// the wait-state rule).  hipcc compiles the portable CIOS to 128 mad + 128
// `v_lshl_add_u64` + ~370 `v_mov` (zero-extension of 32-bit addends into
// register pairs): ~660 instructions; this form is ~330.
SNARKV_HD Fq fq_mul(const Fq& a, const Fq& b) {
  constexpr uint32_t kP0 = fq_p(0), kP1 = fq_p(1), kP2 = fq_p(2), kP3 = fq_p(3), kP4 = fq_p(4), kP5 = fq_p(5),
                     kP6 = fq_p(6), kP7 = fq_p(7);
#include "fq_mul_asm.inc"
  fq_reduce_once(t);
  Fq r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  return r;
}
#else
SNARKV_HD Fq fq_mul(const Fq& a, const Fq& b) { return fq_mul_portable(a, b); }
#endif

SNARKV_HD Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }

// canonical 32-byte little-endian  <->  Montgomery
SNARKV_HD Fq fq_from_canonical(const uint32_t w[8]) {
  constexpr uint32_t r2[8] = SNARKV_FQ_R2_MONT;
  Fq a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a.v[i] = w[i];
    b.v[i] = r2[i];
  }
  return fq_mul(a, b);
}

SNARKV_HD void fq_to_canonical(const Fq& a, uint32_t w[8]) {
  Fq one;
#pragma unroll
  for (int i = 0; i < 8; ++i) one.v[i] = (i == 0) ? 1u : 0u;
  Fq r = fq_mul(a, one);
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = r.v[i];
}

// true iff w (canonical integer, 8 LE words) < p
SNARKV_HD bool fq_canonical_in_range(const uint32_t w[8]) {
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t x = (uint64_t)w[i] - fq_p(i) - borrow;
    borrow = (x >> 32) & 1u;
  }
  return borrow != 0;
}

// a^e for a 256-bit exponent given as 8 LE words (lane-uniform control flow).
SNARKV_HD_NOINLINE Fq fq_pow(const Fq& a, const uint32_t* e) {
  Fq res = fq_one();
  for (int i = 7; i >= 0; --i) {
    uint32_t w = e[i];
    for (int b = 31; b >= 0; --b) {
      res = fq_sqr(res);
      if ((w >> b) & 1u) res = fq_mul(res, a);
    }
  }
  return res;
}

// a^(p-2): Fermat inversion, fixed (lane-uniform) exponent.  inv(0) = 0.
// Not inlined: one copy per kernel keeps code size in check.
SNARKV_HD_NOINLINE Fq fq_inv(const Fq& a) {
  constexpr uint32_t e[8] = SNARKV_FQ_P_MINUS_2_LIMBS;
  Fq res = fq_one();
  for (int i = 7; i >= 0; --i) {
    uint32_t w = e[i];
    for (int b = 31; b >= 0; --b) {
      res = fq_sqr(res);
      if ((w >> b) & 1u) res = fq_mul(res, a);
    }
  }
  return res;
}

}  // namespace snarkv
