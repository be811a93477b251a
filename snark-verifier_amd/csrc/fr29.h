// BN254 SCALAR field Fr on the lazy 9x29-bit signed-limb form (Montgomery, R = 2^261):
// the arithmetic of the Poseidon transcript kernel (poseidon.hip).  Same
// algorithms as fq29.h (one 64-bit column accumulator, one mad per partial
// product, limb-wise add/sub with lazy carries); only what Poseidon needs.
#pragma once
#include "fq29.h"

namespace snarkv {

struct Fr29 {
  int32_t v[9];
};

SNARKV_HD int32_t fr29_r(int i) {
  constexpr int32_t rl[9] = SNARKV_FR29_P_LIMBS;
  return rl[i];
}
SNARKV_HD Fr29 fr29_zero() {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = 0;
  return r;
}
SNARKV_HD Fr29 fr29_one() {
  constexpr int32_t o[9] = SNARKV_FR29_ONE_LIMBS;
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = o[i];
  return r;
}
SNARKV_HD Fr29 fr29_add(const Fr29& a, const Fr29& b) {
  Fr29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// carry-normalise: limbs 0..7 -> [0, 2^29), limb 8 keeps the sign; value unchanged
SNARKV_HD Fr29 fr29_norm(const Fr29& a) {
  Fr29 r;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int32_t t = a.v[i] + c;
    r.v[i] = t & kMask29;
    c = t >> 29;
  }
  r.v[8] = a.v[8] + c;
  return r;
}
// Montgomery product a*b*2^-261 (mod r); |limb| < 2^30 on both sides.  (Plain C products on purpose: the
// explicit `fq29_smad` form that pays off in the throughput-bound G1 kernels slows the latency-bound
// Poseidon kernel down -- one wavefront per SIMD, where the compiler's two interleaved chains hide latency.)  Result carry-normalised,
// value in (-r/8, 9r/8) for |a|, |b| < 8r.
SNARKV_HD Fr29 fr29_mul(const Fr29& a, const Fr29& b) {
  int32_t m[9];
  Fr29 r;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * fr29_r(k - i);
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)SNARKV_FR29_NINV) & (uint32_t)kMask29);
    acc += (int64_t)m[k] * fr29_r(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)m[i] * fr29_r(k - i);
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}
// x^5
SNARKV_HD Fr29 fr29_pow5(const Fr29& x) {
  Fr29 x2 = fr29_mul(x, x);
  Fr29 x4 = fr29_mul(x2, x2);
  return fr29_mul(x4, x);
}
// unique representative in [0, r), carry-normalised, still in the Montgomery domain; |x| < 8r
SNARKV_HD Fr29 fr29_canon_residue(const Fr29& x) {
  Fr29 y = fr29_mul(fr29_norm(x), fr29_one());
  Fr29 t;
  int32_t neg = y.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = y.v[i] + (fr29_r(i) & neg);
  t = fr29_norm(t);
  Fr29 d;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int32_t s = t.v[i] - fr29_r(i) + c;
    d.v[i] = s & kMask29;
    c = s >> 29;
  }
  d.v[8] = t.v[8] - fr29_r(8) + c;
  int32_t keep = d.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = (t.v[i] & keep) | (d.v[i] & ~keep);
  return t;
}
// 8 x u32 canonical integer (< r) -> Montgomery limbs
SNARKV_HD Fr29 fr29_from_canonical(const uint32_t w[8]) {
  Fr29 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 29 * i;
    int word = bit >> 5, sh = bit & 31;
    uint64_t v = w[word];
    if (word + 1 < 8) v |= (uint64_t)w[word + 1] << 32;
    a.v[i] = (int32_t)((uint32_t)(v >> sh) & (uint32_t)kMask29);
  }
  constexpr int32_t r2[9] = SNARKV_FR29_R2_LIMBS;
  Fr29 b;
#pragma unroll
  for (int i = 0; i < 9; ++i) b.v[i] = r2[i];
  return fr29_mul(a, b);
}
// Montgomery limbs (|a| < 8r) -> canonical integer words
SNARKV_HD void fr29_to_canonical(const Fr29& a, uint32_t w[8]) {
  Fr29 one_raw = fr29_zero();
  one_raw.v[0] = 1;
  Fr29 y = fr29_mul(fr29_norm(a), one_raw);  // a * R^-1
  Fr29 t;
  int32_t neg = y.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = y.v[i] + (fr29_r(i) & neg);
  t = fr29_norm(t);
  Fr29 d;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int32_t s = t.v[i] - fr29_r(i) + c;
    d.v[i] = s & kMask29;
    c = s >> 29;
  }
  d.v[8] = t.v[8] - fr29_r(8) + c;
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

// halo2curves' in-memory Fr (the four u64 limbs of a * 2^256 mod r) -> the canonical integer a: one Montgomery product by
// 2^5 (x * 32 * 2^-261 = x * 2^-256), then the conditional +r / -r of fr29_to_canonical
SNARKV_HD void fr_words_from_mont256(const uint32_t in[8], uint32_t out[8]) {
  Fr29 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 29 * i;
    int word = bit >> 5, sh = bit & 31;
    uint64_t v = in[word];
    if (word + 1 < 8) v |= (uint64_t)in[word + 1] << 32;
    a.v[i] = (int32_t)((uint32_t)(v >> sh) & (uint32_t)kMask29);
  }
  Fr29 m = fr29_zero();
  m.v[0] = 32;
  Fr29 y = fr29_mul(a, m);
  Fr29 t;
  int32_t neg = y.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = y.v[i] + (fr29_r(i) & neg);
  t = fr29_norm(t);
  Fr29 d;
  int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int32_t s = t.v[i] - fr29_r(i) + c;
    d.v[i] = s & kMask29;
    c = s >> 29;
  }
  d.v[8] = t.v[8] - fr29_r(8) + c;
  int32_t keep = d.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) t.v[i] = (t.v[i] & keep) | (d.v[i] & ~keep);
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 29 * i;
    int word = bit >> 5, sh = bit & 31;
    uint64_t v = (uint64_t)(uint32_t)t.v[i] << sh;
    out[word] |= (uint32_t)v;
    if (word + 1 < 8) out[word + 1] |= (uint32_t)(v >> 32);
  }
}

}  // namespace snarkv
