// GLV scalar decomposition for G1 of a j = 0 curve (BN254, pallas: constants from gen_consts.py).
//
// Such a curve has the efficient endomorphism phi(x, y) = (beta x, y) = lambda (x, y).
// Writing k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^127 turns an n-point
// MSM with 254-bit scalars into a 2n-point MSM with 127-bit scalars: the same
// number of bucket additions, but HALF the windows -- half the buckets to
// reduce and half the 2^(c w) doubling chain, which is the part of the
// reference's algorithm (the `result.double()` loop, snark-verifier/src/util/
// msm.rs:285-287) that no amount of parallel hardware shortens.  The result is
// the same group element, hence the same canonical bytes.
//
//   c1 = round(k b2 / r), c2 = round(-k b1 / r)   (via g_i = floor(2^288 |b|/r): error < 2^-34)
//   k1 = k - c1 a1 - c2 a2,   k2 = -c1 b1 - c2 b2
// Any integers c1, c2 give an exact identity k1 + k2 lambda = k (mod r); the
// rounding only bounds the size: |k1| <= (1/2 + 2^-34)(a1 + a2), |k2| <= (1/2 + 2^-34)(|b1| + b2), both
// < 2^127 for BN254 and pallas (asserted by gen_consts.py).
#pragma once
#include "fq.h"

namespace snarkv {

// out[0..nout) = low nout words of a (na words) * b (nb words)
template <int NA, int NB, int NOUT>
SNARKV_HD void mp_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
#pragma unroll
  for (int i = 0; i < NOUT; ++i) out[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (i + j < NOUT) {
        uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
        out[i + j] = (uint32_t)t;
        carry = t >> 32;
      }
    }
    if (i + NB < NOUT) out[i + NB] = (uint32_t)carry;  // slot untouched so far in this row
  }
}

// k: canonical scalar (8 LE words, < r).  out[0..3] = |k1|, out[4..7] = |k2|
// (each < 2^127); bit 127 of each half carries its sign (1 = negative).
SNARKV_HD void glv_decompose(const uint32_t k[8], uint32_t out[8]) {
  // uniform widths for every curve (gen_consts.py `glv_section`): lattice entries 4 words, g_i 6 words
  constexpr uint32_t A1[4] = SNARKV_GLV_A1;
  constexpr uint32_t NB1[4] = SNARKV_GLV_NEG_B1;
  constexpr uint32_t A2[4] = SNARKV_GLV_A2;
  constexpr uint32_t B2[4] = SNARKV_GLV_B2;
  constexpr uint32_t G1[6] = SNARKV_GLV_G1;
  constexpr uint32_t G2[6] = SNARKV_GLV_G2;
  uint32_t a1[4], nb1[4], a2[4], b2[4], g1[6], g2[6];
  for (int i = 0; i < 4; ++i) { a1[i] = A1[i]; nb1[i] = NB1[i]; a2[i] = A2[i]; b2[i] = B2[i]; }
  for (int i = 0; i < 6; ++i) { g1[i] = G1[i]; g2[i] = G2[i]; }

  // c_i = (k*g_i + 2^287) >> 288  (< 2^131): round(k |b| / r) to within 2^-34
  uint32_t t1[14], t2[14];
  mp_mul<8, 6, 14>(k, g1, t1);
  mp_mul<8, 6, 14>(k, g2, t2);
  uint32_t c1[5], c2[5];
  {
    uint64_t carry = ((uint64_t)t1[8] + 0x80000000u) >> 32;
    for (int i = 0; i < 5; ++i) {
      uint64_t v = (uint64_t)t1[9 + i] + carry;
      c1[i] = (uint32_t)v;
      carry = v >> 32;
    }
    carry = ((uint64_t)t2[8] + 0x80000000u) >> 32;
    for (int i = 0; i < 5; ++i) {
      uint64_t v = (uint64_t)t2[9 + i] + carry;
      c2[i] = (uint32_t)v;
      carry = v >> 32;
    }
  }
  // everything below modulo 2^160 (two's complement): the results are < 2^127 in magnitude
  uint32_t p1[5], p2[5], k1[5], k2[5];
  mp_mul<5, 4, 5>(c1, a1, p1);   // c1*a1
  mp_mul<5, 4, 5>(c2, a2, p2);   // c2*a2
  {
    uint64_t borrow = 0;
    for (int i = 0; i < 5; ++i) {
      uint64_t v = (uint64_t)k[i] - p1[i] - borrow;
      k1[i] = (uint32_t)v;
      borrow = (v >> 32) & 1u;
    }
    borrow = 0;
    for (int i = 0; i < 5; ++i) {
      uint64_t v = (uint64_t)k1[i] - p2[i] - borrow;
      k1[i] = (uint32_t)v;
      borrow = (v >> 32) & 1u;
    }
  }
  mp_mul<5, 4, 5>(c1, nb1, p1);  // c1*|b1|  (= -c1*b1)
  mp_mul<5, 4, 5>(c2, b2, p2);   // c2*b2
  {
    uint64_t borrow = 0;
    for (int i = 0; i < 5; ++i) {
      uint64_t v = (uint64_t)p1[i] - p2[i] - borrow;
      k2[i] = (uint32_t)v;
      borrow = (v >> 32) & 1u;
    }
  }
  // magnitude + sign
  uint32_t* halves[2] = {k1, k2};
  for (int h = 0; h < 2; ++h) {
    uint32_t* v = halves[h];
    uint32_t neg = v[4] >> 31;
    if (neg) {
      uint64_t carry = 1;
      for (int i = 0; i < 5; ++i) {
        uint64_t x = (uint64_t)(~v[i]) + carry;
        v[i] = (uint32_t)x;
        carry = x >> 32;
      }
    }
    for (int i = 0; i < 4; ++i) out[4 * h + i] = v[i];
    out[4 * h + 3] |= neg << 31;  // |k| < 2^127: bit 127 is free
  }
}

}  // namespace snarkv
