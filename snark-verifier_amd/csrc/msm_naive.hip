// Batched small-MSM kernels: the device side of
// `NativeLoader::multi_scalar_multiplication`
// (reference snark-verifier/src/loader/native.rs:61-71).
//
// The accumulation path produces MANY independent small MSMs (per proof a
// ~21-term and a ~3-term one from Gwc19/Bdfg21::verify, then two (m+1)-term
// ones from KzgAs::verify -- SURVEY.md section 0 item 6), so the launch unit
// is a SEGMENTED MSM:
//   K1  k_term_scalar_mul : one lane per (term, GLV half): k = k1 + k2*lambda
//                           splits `*base * scalar` (native.rs:67) into two
//                           independent 127-step double-and-add chains on P and
//                           phi(P), on the lazy 9x29-bit field
//   K2  k_segment_fold    : one wave per MSM folds its partials (`reduce(|a,v| a+v)`,
//                           native.rs:68), then `to_affine()` (native.rs:70) and
//                           canonical little-endian store.
// A single small MSM is latency-bound by construction (one dependency chain);
// the segmented launch is what fills the machine.
#include "ctx.hpp"
#include "g1.h"
#include "g1_29.h"
#include "glv.h"
#include "fr29.h"
#include <algorithm>

namespace snarkv {

__device__ __forceinline__ void load_words16(const uint32_t* __restrict__ src, uint32_t* dst, int n16) {
  const uint4* s = reinterpret_cast<const uint4*>(src);
#pragma unroll
  for (int i = 0; i < n16; ++i) {
    uint4 v = s[i];
    dst[4 * i + 0] = v.x;
    dst[4 * i + 1] = v.y;
    dst[4 * i + 2] = v.z;
    dst[4 * i + 3] = v.w;
  }
}

// One (scalar, base) term as the boundary hands it over: canonical integers (the wire form), or -- `mont` -- halo2curves'
// in-memory form, a * 2^256 mod r resp. mod p in 4 x u64 (SNARKV_FLAG_MONTGOMERY).  The point enters the 9 x 29-bit
// Montgomery domain by ONE product either way (another constant); the scalar costs one Fr product more when `mont`.
__device__ __forceinline__ G1Affine29 load_term(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ points,
                                                size_t t, uint32_t mont, uint32_t (&k)[8]) {
  uint32_t pw[16];
  load_words16(scalars + t * 8, k, 2);
  load_words16(points + t * 16, pw, 4);
  if (mont) {
    uint32_t c[8];
    fr_words_from_mont256(k, c);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = c[j];
  }
  return g1a29_from_words(pw, mont != 0);
}

// |k| * Q for a 127-bit magnitude: left-to-right double-and-add on the lazy
// 9x29-bit field.  FAST: branch-free adders, the caller checks the degenerate
// flag; CAREFUL: explicit exceptional cases.
template <bool CAREFUL>
__device__ __forceinline__ G1Xyzz29 half_scalar_mul(const G1Affine29& q, const uint32_t k[4]) {
  G1Xyzz29 acc = xyzz29_identity();
  bool started = false;
  for (int i = 3; i >= 0; --i) {
    uint32_t w = k[i];
    for (int b = 31; b >= 0; --b) {
      if (started) {
        if (!CAREFUL || !xyzz29_is_identity(acc)) acc = xyzz29_double(acc);
      }
      if ((w >> b) & 1u) {
        if (!started) {
          acc = xyzz29_from_affine(q);
          started = true;
        } else if (CAREFUL) {
          xyzz29_madd_careful(acc, q);
        } else {
          xyzz29_madd_fast(acc, q);
        }
      }
    }
  }
  return acc;
}

// |k| * Q with a FIXED 3-bit window and signed digits, every lane of the wavefront in the same step:
//   table    2Q, 3Q, 4Q (XYZZ) in a global scratch, one column per lane (limb-major: every access one segment; it stays in
//            L2) -- not in LDS: 27 KB per wavefront there capped a CU at five wavefronts, and with several launches in
//            flight the kernel's throughput fell back to the bit-serial form's; Q itself stays in registers
//   digits   |k| = sum d_i 8^i, d_i in [-3, 4] (a digit above 4 becomes d - 8 with a carry up), 43 of them for 127 bits,
//            recoded low to high into LDS bytes, consumed high to low
//   step     acc <- 8 acc (three doublings; the identity doubles to itself), then acc += sign * T[|d|] by the full XYZZ
//            addition, the operand picked by a per-lane LDS read
// The bit-serial form above costs a WAVEFRONT 127 doublings + 127 mixed additions (some lane's bit is always set, so
// the addition is issued at every step): ~2 400 field products.  This one: 3 + 129 doublings + 43 additions ~ 1 800,
// and no lane waits for another's branch.  A fast addition can only meet P = +-Q here while acc is still the identity
// (8 x prefix >= 8 > |d|), which `started` covers; the degenerate check + careful bit-serial redo stay as the net.
#ifndef SNARKV_NAIVE_WINDOW
#define SNARKV_NAIVE_WINDOW 1  // 0: the bit-serial double-and-add (A/B: profiles/r03_ab_naive_window.txt)
#endif
#ifndef SNARKV_NAIVE_WAVES
#define SNARKV_NAIVE_WAVES 2  // wavefronts per SIMD k_term_scalar_mul is compiled for (3: 168 VGPRs + 48 B of spills)
#endif
constexpr int kWinDigits = 43;  // ceil(127 / 3) + the carry into the one-bit top digit

__device__ __forceinline__ G1Xyzz29 half_scalar_mul_w3(const G1Affine29& q, const uint32_t k[4], int32_t* __restrict__ tabg,
                                                       int8_t (*dig)[64]) {
  const uint32_t lane = threadIdx.x;
  // this wavefront's slice of the scratch: [entry 0..2][limb 0..35][lane]
  int32_t(*tab)[36][64] = reinterpret_cast<int32_t(*)[36][64]>(tabg + (size_t)blockIdx.x * 3 * 36 * 64);
  // table: 2Q, 3Q, 4Q
  G1Xyzz29 t2 = xyzz29_double_affine(q), t3 = t2;
  xyzz29_madd_fast(t3, q);
  G1Xyzz29 t4 = xyzz29_double(t2);
  auto put = [&](int e, const G1Xyzz29& v) {
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      tab[e][l][lane] = v.x.v[l];
      tab[e][9 + l][lane] = v.y.v[l];
      tab[e][18 + l][lane] = v.zz.v[l];
      tab[e][27 + l][lane] = v.zzz.v[l];
    }
  };
  put(0, t2);
  put(1, t3);
  put(2, t4);
  // signed digits, low to high
  uint32_t carry = 0;
  for (int i = 0; i < kWinDigits; ++i) {
    const int bit = 3 * i, word = bit >> 5, sh = bit & 31;
    uint32_t w0 = word == 0 ? k[0] : word == 1 ? k[1] : word == 2 ? k[2] : word == 3 ? k[3] : 0u;
    uint32_t w1 = word == 0 ? k[1] : word == 1 ? k[2] : word == 2 ? k[3] : 0u;
    uint32_t raw = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh) & 7u;
    raw += carry;
    carry = raw > 4u ? 1u : 0u;
    dig[i][lane] = (int8_t)((int)raw - (carry ? 8 : 0));
  }
  // (the lane reads back only what it wrote: no barrier; the compiler keeps LDS accesses of one lane in order)
  G1Xyzz29 acc = xyzz29_identity();
  bool started = false;
#pragma unroll 1
  for (int i = kWinDigits - 1; i >= 0; --i) {
    acc = xyzz29_double(xyzz29_double(xyzz29_double(acc)));  // 8 acc (all-zero stays all-zero)
    const int d = dig[i][lane];
    const int a = d < 0 ? -d : d;
    if (a != 0) {
      G1Xyzz29 sel;
      if (a == 1) {
        sel = xyzz29_from_affine(q);
      } else {
#pragma unroll
        for (int l = 0; l < 9; ++l) {
          sel.x.v[l] = tab[a - 2][l][lane];
          sel.y.v[l] = tab[a - 2][9 + l][lane];
          sel.zz.v[l] = tab[a - 2][18 + l][lane];
          sel.zzz.v[l] = tab[a - 2][27 + l][lane];
        }
      }
      if (d < 0) sel.y = fq29_neg(sel.y);
      G1Xyzz29 sum = acc;
      xyzz29_add_fast(sum, sel);
      acc = started ? sum : sel;
      started = true;
    }
  }
  return acc;
}

// K1: one lane per (term, GLV half).  k = k1 + k2*lambda with |k_i| < 2^127
// (glv.h) turns `*base * scalar` (reference native.rs:67), a 254-step chain,
// into two independent 127-step chains on P and phi(P) = (beta x, y).
__global__ void __launch_bounds__(64, SNARKV_NAIVE_WAVES) k_term_scalar_mul(const uint32_t* __restrict__ scalars,
                                                         const uint32_t* __restrict__ points,
                                                         G1Xyzz29* __restrict__ out, uint32_t n_terms,
                                                         int32_t* __restrict__ tabg, uint32_t mont) {
#if SNARKV_NAIVE_WINDOW
  __shared__ int8_t dig[kWinDigits][64];
#endif
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= 2 * n_terms) return;
  uint32_t t = g >> 1, h = g & 1u;
  uint32_t k[8], halves[8];
  G1Affine29 q = load_term(scalars, points, t, mont, k);
  glv_decompose(k, halves);
  uint32_t mag[4] = {halves[4 * h], halves[4 * h + 1], halves[4 * h + 2], halves[4 * h + 3] & 0x7FFFFFFFu};
  uint32_t neg = halves[4 * h + 3] >> 31;
  if (g1a29_is_identity(q) || (mag[0] | mag[1] | mag[2] | mag[3]) == 0) {
    out[g] = xyzz29_identity();
    return;
  }
  if (h) {
    constexpr int32_t bl[9] = SNARKV_GLV_BETA29_LIMBS;
    Fq29 beta;
#pragma unroll
    for (int j = 0; j < 9; ++j) beta.v[j] = bl[j];
    q.x = fq29_canon_residue(fq29_mul(q.x, beta));
  }
  if (neg) q.y = fq29_neg(q.y);
#if SNARKV_NAIVE_WINDOW
  G1Xyzz29 r = half_scalar_mul_w3(q, mag, tabg, dig);
#else
  G1Xyzz29 r = half_scalar_mul<false>(q, mag);
#endif
  if (xyzz29_is_degenerate(r)) {  // P = +-Q met on the way (or a true identity): redo carefully
    r = half_scalar_mul<true>(q, mag);
    if (!xyzz29_is_identity(r) && xyzz29_is_degenerate(r)) r = xyzz29_identity();
  }
  out[g] = r;
}

// K1 for THROUGHPUT-bound launches: one lane per GROUP of up to kGroupMax terms OF ONE SEGMENT, both GLV halves of every
// term, ONE accumulator and one doubling chain for all of them (Straus), and AFFINE tables so that every addition is mixed.
// The two-lane form above spends 129 doublings (~1 160 products) per half-scalar next to 43 full additions; here a term
// costs 1 160 / K + 2 x 43 x 7/8 mixed additions (~750) + ~80 for its table: K = 3: ~1 220 products against ~3 600 -- on a
// chain K times as long per step, so only for launches that fill the machine (tens of thousands of terms, or several
// launches in flight: the context's throughput hint).  Round 3 had the K = 1 case with projective tables (one lane per
// term: 2 360 products); measured steps: profiles/r04_ab_naive_group.txt.
//   lanes     segment s of L_s terms gets ceil(L_s / K) lanes; K in {2, 3, 4} is chosen ON THE DEVICE (the offsets live
//             there) as the shortest by  passes of the resident grid x (1 160 + K x 830): three small kernels count, choose + scan, fill the
//             lane -> segment map, and a lane finds its segment by binary search
//   grid      at most the machine's resident wavefronts; a block walks its share of the lanes, so the table scratch is
//             sized by the machine, not by the launch
//   tables    per term Q, 2Q, 3Q, 4Q as AFFINE points (x, y, beta x) in the global scratch, limb-major per lane: 2Q .. 4Q of
//             all the lane's terms are brought to affine by ONE field inversion (Montgomery's trick over <= 12 elements: 8
//             products each + an inversion worth ~55), which turns 14-product additions into 10-product ones
//   digits    signed 3-bit, both halves, as NIBBLES in LDS (K x 2 x 22 bytes per lane)
// Output: the group's sum in the slot of its first term, the identity in the group's other slots (the fold is unchanged).
// A degenerate accumulator (P = +-Q met: e.g. a base listed twice in one group) sends every term of the group through the
// careful bit-serial form.  A fast addition meets P = +-Q otherwise only if  sum_j (u_j + v_j lambda) P_j = +-d P_i  for
// prefixes u, v of the lane's scalars: for independent bases never, for crafted ones the degenerate flag is the net.
constexpr int kGroupMax = 4;
constexpr int kGroupRows = 27 + 3 * 45 + 9;  // table rows per term: Q's x, y, beta x; 2Q, 3Q, 4Q with 45 rows each while they are
                                             // built (X, Y, ZZ, ZZZ, prefix product; x, y, beta x afterwards); the prefix before the term

__global__ void __launch_bounds__(256) k_gmap_count(const uint32_t* __restrict__ offsets, uint32_t n_msm, uint32_t* __restrict__ bsum) {
  __shared__ uint32_t sh[3][256];
  const uint32_t tid = threadIdx.x, nblk = gridDim.x;
  uint32_t c[3] = {0, 0, 0};
  for (uint32_t q = 0; q < 4; ++q) {
    const uint32_t sg = blockIdx.x * 1024 + q * 256 + tid;
    if (sg < n_msm) {
      const uint32_t L = offsets[sg + 1] - offsets[sg];
      c[0] += (L + 1) / 2, c[1] += (L + 2) / 3, c[2] += (L + 3) / 4;
    }
  }
  for (int k = 0; k < 3; ++k) sh[k][tid] = c[k];
  __syncthreads();
  for (uint32_t st = 128; st >= 1; st >>= 1) {
    if (tid < st)
      for (int k = 0; k < 3; ++k) sh[k][tid] += sh[k][tid + st];
    __syncthreads();
  }
  if (tid < 3) bsum[tid * nblk + blockIdx.x] = sh[tid][0];
}

// one block: totals per K, the choice, and the exclusive scan of the chosen K's block sums (in place in its row)
__global__ void __launch_bounds__(1024) k_gmap_choose(uint32_t* __restrict__ bsum, uint32_t nblk, uint32_t* __restrict__ choice,
                                                      uint32_t force_k, uint32_t slots) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t tot[3], run, kk;
  const uint32_t tid = threadIdx.x;
  for (int k = 0; k < 3; ++k) {
    uint32_t a = 0;
    for (uint32_t b = tid; b < nblk; b += 1024) a += bsum[k * nblk + b];
    sh[tid] = a;
    __syncthreads();
    for (uint32_t st = 512; st >= 1; st >>= 1) {
      if (tid < st) sh[tid] += sh[tid + st];
      __syncthreads();
    }
    if (tid == 0) tot[k] = sh[0];
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t best = 0;
    unsigned long long bc = ~0ull;
    uint32_t bt = ~0u;
    for (uint32_t k = 0; k < 3; ++k) {
      // duration ~ (passes of the resident grid over the lanes) x (one lane's chain): a launch that does not fill the
      // machine is as long as its chain, whatever its total work (profiles/r05_ab_fixed_base.txt: 16 x 1 024 proofs with
      // 12 + 3-term segments took K = 4 by total work and ran 4.3 ms on one wavefront per SIMD; K = 2 fills the same slots
      // with a chain 0.63x as long: 3.0 ms).  Ties go to the smaller total.
      const unsigned long long waves = ((unsigned long long)tot[k] + 63) / 64;
      const unsigned long long passes = slots ? (waves + slots - 1) / slots : 1;
      const unsigned long long cost = (passes ? passes : 1) * (1160ull + (k + 2) * 830ull);
      // (cost, total work) compared lexicographically: `tot` is a lane count that can exceed any fixed weight of the
      // primary term, so it is never packed into it
      if (cost < bc || (cost == bc && tot[k] < bt)) bc = cost, bt = tot[k], best = k;
    }
    if (force_k >= 2 && force_k <= 4) best = force_k - 2;
    kk = best;
    run = 0;
    choice[0] = best + 2;
    choice[1] = tot[best];
  }
  __syncthreads();
  uint32_t* row = bsum + kk * nblk;
  for (uint32_t base = 0; base < nblk; base += 1024) {
    const uint32_t idx = base + tid;
    const uint32_t v = idx < nblk ? row[idx] : 0;
    sh[tid] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      const uint32_t t = tid >= off ? sh[tid - off] : 0;
      __syncthreads();
      sh[tid] += t;
      __syncthreads();
    }
    const uint32_t r0 = run;
    if (idx < nblk) row[idx] = r0 + sh[tid] - v;
    __syncthreads();
    if (tid == 1023) run = r0 + sh[1023];
    __syncthreads();
  }
}

// base[s] = lanes before segment s for the chosen K; base[n_msm] = all of them
__global__ void __launch_bounds__(256) k_gmap_fill(const uint32_t* __restrict__ offsets, uint32_t n_msm, const uint32_t* __restrict__ bsum,
                                                   const uint32_t* __restrict__ choice, uint32_t* __restrict__ base) {
  __shared__ uint32_t sh[256];
  const uint32_t tid = threadIdx.x, nblk = gridDim.x, K = choice[0];
  uint32_t c[4], mine = 0;
  for (uint32_t q = 0; q < 4; ++q) {
    const uint32_t sg = blockIdx.x * 1024 + tid * 4 + q;
    c[q] = sg < n_msm ? (offsets[sg + 1] - offsets[sg] + K - 1) / K : 0;
    mine += c[q];
  }
  sh[tid] = mine;
  __syncthreads();
  for (uint32_t off = 1; off < 256; off <<= 1) {
    const uint32_t t = tid >= off ? sh[tid - off] : 0;
    __syncthreads();
    sh[tid] += t;
    __syncthreads();
  }
  uint32_t run = bsum[(K - 2) * nblk + blockIdx.x] + sh[tid] - mine;
  for (uint32_t q = 0; q < 4; ++q) {
    const uint32_t sg = blockIdx.x * 1024 + tid * 4 + q;
    if (sg < n_msm) base[sg] = run;
    run += c[q];
  }
  if (blockIdx.x == 0 && tid == 0) base[n_msm] = choice[1];
}

__global__ void __launch_bounds__(64, SNARKV_NAIVE_WAVES)
    k_term_scalar_mul_group(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ points, G1Xyzz29* __restrict__ out,
                            const uint32_t* __restrict__ offsets, uint32_t n_msm, const uint32_t* __restrict__ base,
                            const uint32_t* __restrict__ choice, int32_t* __restrict__ tabg, uint32_t mont) {
  __shared__ uint8_t dig[kGroupMax][2][(kWinDigits + 1) / 2][64];
  const uint32_t lane = threadIdx.x, K = choice[0], T = choice[1];
  int32_t(*tab)[kGroupRows][64] = reinterpret_cast<int32_t(*)[kGroupRows][64]>(tabg + (size_t)blockIdx.x * kGroupMax * kGroupRows * 64);
  constexpr int32_t bl[9] = SNARKV_GLV_BETA29_LIMBS;
  Fq29 beta;
#pragma unroll
  for (int j = 0; j < 9; ++j) beta.v[j] = bl[j];
#pragma unroll 1
  for (uint32_t g0 = blockIdx.x * 64; g0 < T; g0 += gridDim.x * 64) {
    const uint32_t g = g0 + lane;
    const bool live = g < T;
    uint32_t first = 0, cnt = 0;
    if (live) {  // the segment whose lanes include g: the last s with base[s] <= g
      uint32_t lo = 0, hi = n_msm;  // base[lo] <= g < base[hi]
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (base[mid] <= g) lo = mid;
        else hi = mid;
      }
      first = offsets[lo] + (g - base[lo]) * K;
      const uint32_t end = offsets[lo + 1];
      cnt = end - first < K ? end - first : K;
    }
    // ---- per term: Q and the XYZZ forms of 2Q, 3Q, 4Q into the scratch, the running product of their ZZ ZZZ; digits
    uint32_t hasm = 0;
    Fq29 running = fq29_one();
#pragma unroll 1
    for (uint32_t j = 0; j < K; ++j) {
      bool has = live && j < cnt;
      uint32_t mag[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, neg[2] = {0, 0};
      if (has) {
        uint32_t k[8], halves[8];
        const G1Affine29 q = load_term(scalars, points, first + j, mont, k);
        glv_decompose(k, halves);
        uint32_t any = 0;
        for (int h = 0; h < 2; ++h) {
          for (int w = 0; w < 4; ++w) mag[h][w] = halves[4 * h + w];
          neg[h] = mag[h][3] >> 31;
          mag[h][3] &= 0x7FFFFFFFu;
          any |= mag[h][0] | mag[h][1] | mag[h][2] | mag[h][3];
        }
        if (g1a29_is_identity(q) || any == 0) {
          has = false;  // contributes nothing: all digits zero, table untouched
        } else {
          hasm |= 1u << j;
          const Fq29 qbx = fq29_mul(q.x, beta);
#pragma unroll
          for (int l = 0; l < 9; ++l) {
            tab[j][l][lane] = q.x.v[l];
            tab[j][9 + l][lane] = q.y.v[l];
            tab[j][18 + l][lane] = qbx.v[l];
            tab[j][162 + l][lane] = running.v[l];  // the prefix product before this term's three elements
          }
          G1Xyzz29 t2 = xyzz29_double_affine(q), t3 = t2;
          xyzz29_madd_fast(t3, q);
          G1Xyzz29 t4 = xyzz29_double(t2);
          auto put = [&](int e, const G1Xyzz29& v) {
            running = fq29_mul(running, fq29_mul(v.zz, v.zzz));
            const int r0 = 27 + 45 * e;
#pragma unroll
            for (int l = 0; l < 9; ++l) {
              tab[j][r0 + l][lane] = v.x.v[l];
              tab[j][r0 + 9 + l][lane] = v.y.v[l];
              tab[j][r0 + 18 + l][lane] = v.zz.v[l];
              tab[j][r0 + 27 + l][lane] = v.zzz.v[l];
              tab[j][r0 + 36 + l][lane] = running.v[l];  // ... including this element
            }
          };
          put(0, t2);
          put(1, t3);
          put(2, t4);
        }
      }
      for (int h = 0; h < 2; ++h) {
        uint32_t carry = 0, pend = 0;
        for (int i = 0; i < kWinDigits; ++i) {
          const int bit = 3 * i, word = bit >> 5, sh = bit & 31;
          uint32_t w0 = word == 0 ? mag[h][0] : word == 1 ? mag[h][1] : word == 2 ? mag[h][2] : word == 3 ? mag[h][3] : 0u;
          uint32_t w1 = word == 0 ? mag[h][1] : word == 1 ? mag[h][2] : word == 2 ? mag[h][3] : 0u;
          uint32_t raw = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh) & 7u;
          raw += carry;
          carry = raw > 4u ? 1u : 0u;
          int d = (int)raw - (carry ? 8 : 0);
          if (neg[h]) d = -d;
          const uint32_t enc = has ? (uint32_t)(d + 4) : 4u;  // 4 = the digit 0
          if (i & 1) dig[j][h][i >> 1][lane] = (uint8_t)(pend | (enc << 4));
          else pend = enc;
        }
        dig[j][h][kWinDigits >> 1][lane] = (uint8_t)(pend | (4u << 4));  // 43 digits: the last byte holds one
      }
    }
    // ---- 2Q, 3Q, 4Q of every term to affine: one inversion for the lane, then backwards through the prefix products
    if (hasm) {
      Fq29 inv = fq29_inv(running);  // 1 / (product of all ZZ ZZZ); a doubling or addition of table building never gives ZZ = 0
#pragma unroll 1
      for (int j = (int)K - 1; j >= 0; --j) {
        if (!((hasm >> j) & 1u)) continue;
#pragma unroll 1
        for (int e = 2; e >= 0; --e) {
          const int r0 = 27 + 45 * e, rp = e == 0 ? 162 : r0 - 45 + 36;  // where the prefix BEFORE this element lies
          Fq29 X, Y, ZZ, ZZZ, prev;
#pragma unroll
          for (int l = 0; l < 9; ++l) {
            X.v[l] = tab[j][r0 + l][lane];
            Y.v[l] = tab[j][r0 + 9 + l][lane];
            ZZ.v[l] = tab[j][r0 + 18 + l][lane];
            ZZZ.v[l] = tab[j][r0 + 27 + l][lane];
            prev.v[l] = tab[j][rp + l][lane];
          }
          const Fq29 izn = fq29_mul(inv, prev);   // 1 / (ZZ ZZZ) of this element
          inv = fq29_mul(inv, fq29_mul(ZZ, ZZZ));  // ... and the inverse of the product up to the previous one
          const Fq29 xa = fq29_mul(X, fq29_mul(izn, ZZZ));            // X / ZZ
          const Fq29 ya = fq29_mul(fq29_norm(Y), fq29_mul(izn, ZZ));  // Y / ZZZ
          const Fq29 bxa = fq29_mul(xa, beta);
#pragma unroll
          for (int l = 0; l < 9; ++l) {
            tab[j][r0 + l][lane] = xa.v[l];
            tab[j][r0 + 9 + l][lane] = ya.v[l];
            tab[j][r0 + 18 + l][lane] = bxa.v[l];
          }
        }
      }
    }
    // ---- the shared chain: acc <- 8 acc, then one MIXED addition per non-zero digit of every (term, half)
    G1Xyzz29 acc = xyzz29_identity();
    bool started = false;
#pragma unroll 1
    for (int i = kWinDigits - 1; i >= 0; --i) {
      acc = xyzz29_double(xyzz29_double(xyzz29_double(acc)));  // 8 acc (all-zero stays all-zero)
#pragma unroll 1
      for (uint32_t jh = 0; jh < 2 * K; ++jh) {
        const uint32_t j = jh >> 1, h = jh & 1u;
        const int d = (int)((dig[j][h][i >> 1][lane] >> ((i & 1) * 4)) & 15u) - 4;
        const int a = d < 0 ? -d : d;
        if (a != 0) {
          const int r0 = a == 1 ? 0 : 27 + 45 * (a - 2), xo = h ? 18 : 0;
          G1Affine29 sel;
#pragma unroll
          for (int l = 0; l < 9; ++l) {
            sel.x.v[l] = tab[j][r0 + xo + l][lane];
            sel.y.v[l] = tab[j][r0 + 9 + l][lane];
          }
          if (d < 0) sel.y = fq29_neg(sel.y);
          G1Xyzz29 sum = acc;
          xyzz29_madd_fast(sum, sel);
          acc = started ? sum : xyzz29_from_affine(sel);
          started = true;
        }
      }
    }
    if (!live) continue;
    if (started && xyzz29_is_degenerate(acc)) {  // an exceptional addition on the way: every term of the group, carefully
#pragma unroll 1
      for (uint32_t j = 0; j < cnt; ++j) {
        uint32_t k[8], halves[8], m0[4], m1[4];
        const G1Affine29 q = load_term(scalars, points, first + j, mont, k);
        glv_decompose(k, halves);
        for (int w = 0; w < 4; ++w) m0[w] = halves[w], m1[w] = halves[4 + w];
        const uint32_t n0 = m0[3] >> 31, n1 = m1[3] >> 31;
        m0[3] &= 0x7FFFFFFFu, m1[3] &= 0x7FFFFFFFu;
        G1Xyzz29 r1 = xyzz29_identity(), r2 = xyzz29_identity();
        if (!g1a29_is_identity(q)) {
          G1Affine29 q1 = q, q2 = q;
          q2.x = fq29_canon_residue(fq29_mul(q.x, beta));
          if (n0) q1.y = fq29_neg(q1.y);
          if (n1) q2.y = fq29_neg(q2.y);
          if ((m0[0] | m0[1] | m0[2] | m0[3]) != 0) r1 = half_scalar_mul<true>(q1, m0);
          if (!xyzz29_is_identity(r1) && xyzz29_is_degenerate(r1)) r1 = xyzz29_identity();
          if ((m1[0] | m1[1] | m1[2] | m1[3]) != 0) r2 = half_scalar_mul<true>(q2, m1);
          if (!xyzz29_is_identity(r2) && xyzz29_is_degenerate(r2)) r2 = xyzz29_identity();
        }
        out[2 * (size_t)(first + j)] = r1;
        out[2 * (size_t)(first + j) + 1] = r2;  // the fold adds them with the careful adder
      }
      continue;
    }
    out[2 * (size_t)first] = started ? acc : xyzz29_identity();
    for (uint32_t q = 1; q < 2 * cnt; ++q) out[2 * (size_t)first + q] = xyzz29_identity();
  }
}

// ---- chunked form of K1 for SMALL batches (the machine is far from full) ----
// A 127-step chain per lane leaves >90 % of the SIMDs idle when a job has a
// few thousand terms.  Split every half-scalar into J chunks of 128/J bits:
//   K1a k_term_chain  : lane per (term, half): GLV split, then ONE Jacobian
//                       doubling chain (7 products per doubling instead of the
//                       19 of a double-and-add step) that emits the chunk
//                       bases Q_j = 2^(j*bits) * Q
//   K1b k_term_chunks : lane per (term, half, chunk): (bits)-step double-and-add
//                       on Q_j, then a shuffle tree over the J adjacent lanes
// Dependent field products per term: ~2400 -> ~900 + 23*bits.  Same partial
// layout as K1 (one XYZZ per (term, half)), so K2 is unchanged.
__global__ void __launch_bounds__(64) k_term_chain(const uint32_t* __restrict__ scalars,
                                                    const uint32_t* __restrict__ points,
                                                    G1Xyzz29* __restrict__ chain, uint4* __restrict__ mags,
                                                    uint32_t n_terms, uint32_t J, uint32_t bits, uint32_t mont) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= 2 * n_terms) return;
  uint32_t t = g >> 1, h = g & 1u;
  uint32_t k[8], halves[8];
  G1Affine29 q = load_term(scalars, points, t, mont, k);
  glv_decompose(k, halves);
  uint4 mag = make_uint4(halves[4 * h], halves[4 * h + 1], halves[4 * h + 2], halves[4 * h + 3] & 0x7FFFFFFFu);
  uint32_t neg = halves[4 * h + 3] >> 31;
  if (g1a29_is_identity(q)) mag = make_uint4(0, 0, 0, 0);  // all digits zero -> identity partial
  mags[g] = mag;
  if (h) {
    constexpr int32_t bl[9] = SNARKV_GLV_BETA29_LIMBS;
    Fq29 beta;
#pragma unroll
    for (int j = 0; j < 9; ++j) beta.v[j] = bl[j];
    q.x = fq29_canon_residue(fq29_mul(q.x, beta));
  }
  if (neg) q.y = fq29_norm(fq29_neg(q.y));
  G1Xyzz29* dst = chain + (size_t)g * J;
  Fq29 x = q.x, y = q.y, z = fq29_one();
  dst[0] = jac29_to_xyzz(x, y, z);  // (x, y, 1, 1)
  if ((mag.x | mag.y | mag.z | mag.w) == 0) return;  // bases never read
  for (uint32_t j = 1; j < J; ++j) {
    for (uint32_t i = 0; i < bits; ++i) jac29_double(x, y, z);
    dst[j] = jac29_to_xyzz(x, y, z);
  }
}

// K1a with FOUR lanes per (term, half): the same chain, every doubling by jac29_double_quad (3 dependent products
// instead of 7).  For small jobs -- an aggregation of 64 proofs is 1 666 terms -- the chain IS the launch's duration
// (120 doublings: 0.35 -> 0.19 ms); large jobs keep the one-lane form, whose lanes all do useful work.
__global__ void __launch_bounds__(64) k_term_chain_quad(const uint32_t* __restrict__ scalars,
                                                         const uint32_t* __restrict__ points,
                                                         G1Xyzz29* __restrict__ chain, uint4* __restrict__ mags,
                                                         uint32_t n_terms, uint32_t J, uint32_t bits, uint32_t mont) {
  uint32_t lane4 = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t g = lane4 >> 2, q = lane4 & 3u;  // a quad never straddles a wavefront; all four lanes take the same branches
  if (g >= 2 * n_terms) return;
  uint32_t t = g >> 1, h = g & 1u;
  uint32_t k[8], halves[8];
  G1Affine29 p = load_term(scalars, points, t, mont, k);
  glv_decompose(k, halves);
  uint4 mag = make_uint4(halves[4 * h], halves[4 * h + 1], halves[4 * h + 2], halves[4 * h + 3] & 0x7FFFFFFFu);
  uint32_t neg = halves[4 * h + 3] >> 31;
  if (g1a29_is_identity(p)) mag = make_uint4(0, 0, 0, 0);
  if (q == 0) mags[g] = mag;
  if (h) {
    constexpr int32_t bl[9] = SNARKV_GLV_BETA29_LIMBS;
    Fq29 beta;
#pragma unroll
    for (int j = 0; j < 9; ++j) beta.v[j] = bl[j];
    p.x = fq29_canon_residue(fq29_mul(p.x, beta));
  }
  if (neg) p.y = fq29_norm(fq29_neg(p.y));
  G1Xyzz29* dst = chain + (size_t)g * J;
  Fq29 x = p.x, y = p.y, z = fq29_one();
  if (q == 0) dst[0] = jac29_to_xyzz(x, y, z);
  if ((mag.x | mag.y | mag.z | mag.w) == 0) return;
  for (uint32_t j = 1; j < J; ++j) {
    for (uint32_t i = 0; i < bits; ++i) jac29_double_quad(x, y, z, q);
    G1Xyzz29 b = jac29_to_xyzz(x, y, z);
    if (q == 0) dst[j] = b;
  }
}

__device__ __forceinline__ G1Xyzz29 xyzz29_shfl_xor(const G1Xyzz29& p, int mask) {
  G1Xyzz29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    r.x.v[i] = __shfl_xor(p.x.v[i], mask);
    r.y.v[i] = __shfl_xor(p.y.v[i], mask);
    r.zz.v[i] = __shfl_xor(p.zz.v[i], mask);
    r.zzz.v[i] = __shfl_xor(p.zzz.v[i], mask);
  }
  return r;
}

__device__ __forceinline__ uint32_t mag_bit(const uint4& m, uint32_t b) {
  uint32_t w = (b >> 5) == 0 ? m.x : (b >> 5) == 1 ? m.y : (b >> 5) == 2 ? m.z : m.w;
  return (w >> (b & 31u)) & 1u;
}

__global__ void __launch_bounds__(64) k_term_chunks(const G1Xyzz29* __restrict__ chain,
                                                     const uint4* __restrict__ mags,
                                                     G1Xyzz29* __restrict__ out, uint32_t n_lanes, uint32_t J,
                                                     uint32_t bits) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // J divides 64: a chunk group never straddles a wavefront
  bool live = t < n_lanes;
  uint32_t g = live ? t / J : 0, j = t % J;
  uint4 mag = live ? mags[g] : make_uint4(0, 0, 0, 0);
  G1Xyzz29 acc = xyzz29_identity();
  bool bad = false;
  uint32_t any = 0;
  for (uint32_t i = 0; i < bits; ++i) any |= mag_bit(mag, bits * j + i);
  if (any) {
    G1Xyzz29 base = chain[(size_t)g * J + j];
    base.y = fq29_norm(base.y);
    bool started = false;
    for (int i = (int)bits - 1; i >= 0; --i) {
      if (started) acc = xyzz29_double(acc);
      if (mag_bit(mag, bits * j + (uint32_t)i)) {
        if (!started) {
          acc = base;
          started = true;
        } else {
          // d*Q + Q with 1 < d < 2^bits: never +-Q in a prime-order group
          xyzz29_add_fast(acc, base);
        }
      }
    }
    acc.y = fq29_norm(acc.y);
  }
  // sum the J chunk results of one (term, half): distinct bit ranges of a
  // 127-bit integer, so no two partial sums are equal or opposite
  for (uint32_t s = 1; s < J; s <<= 1) {
    G1Xyzz29 other = xyzz29_shfl_xor(acc, (int)s);
    xyzz29_add_skipid_fast(acc, other, bad);
    acc.y = fq29_norm(acc.y);
    bad = bad || (__shfl_xor((int)bad, (int)s) != 0);
  }
  if (live && j == 0) {
    if (bad || (!xyzz29_is_identity(acc) && xyzz29_is_degenerate(acc))) {  // off-curve input or the like: redo carefully
      G1Xyzz29 b0 = chain[(size_t)g * J];
      G1Affine29 q;
      q.x = b0.x;
      q.y = b0.y;
      uint32_t m4[4] = {mag.x, mag.y, mag.z, mag.w};
      acc = half_scalar_mul<true>(q, m4);
      if (!xyzz29_is_identity(acc) && xyzz29_is_degenerate(acc)) acc = xyzz29_identity();
    }
    out[g] = acc;
  }
}

// K2: G lanes per MSM fold its 2 x terms partials (`reduce(|a,v| a+v)`, native.rs:68), then `to_affine()`
// (native.rs:70).  G = 16 for the ~21-term MSMs of a proof: four MSMs share a wavefront -- 3 strided additions + a
// 4-level tree is as deep as 1 + 6 levels over 64 lanes, in a quarter of the wavefronts, and four lanes instead of one are
// live in the inversion (with many jobs in flight the folds were a quarter of the issued work); G = 64 for longer
// segments; 256 lanes for the (m + 1)-term MSMs of KzgAs::verify, where the strided pre-sum is the critical path.
template <int THREADS, int G>
__global__ void __launch_bounds__(THREADS) k_segment_fold(const G1Xyzz29* __restrict__ parts,
                                                           const uint32_t* __restrict__ offsets,
                                                           uint32_t* __restrict__ out, uint32_t n_msm, uint32_t mont) {
  static_assert(THREADS % G == 0 && (G & (G - 1)) == 0, "G lanes per MSM: a power of two dividing the workgroup");
  __shared__ G1Xyzz29 sh[THREADS];
  const uint32_t tid = threadIdx.x, lane = tid % G;
  const uint32_t k = blockIdx.x * (THREADS / G) + tid / G;
  const bool live = k < n_msm;
  const uint32_t lo = live ? 2 * offsets[k] : 0, hi = live ? 2 * offsets[k + 1] : 0;
  G1Xyzz29 acc = xyzz29_identity();
  for (uint32_t i = lo + lane; i < hi; i += G) xyzz29_add_careful(acc, parts[i]);
  sh[tid] = acc;
  __syncthreads();
  for (uint32_t s = G / 2; s >= 1; s >>= 1) {
    if (lane < s) {
      G1Xyzz29 a = sh[tid];
      xyzz29_add_careful(a, sh[tid + s]);
      sh[tid] = a;
    }
    __syncthreads();
  }
  if (lane == 0 && live) {
    G1Affine29 r = xyzz29_to_affine(sh[tid]);
    uint32_t w[16];
    g1a29_to_words(r, w, mont != 0);
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)k * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
  }
}

// Optional input validation (SNARKV_FLAG_VALIDATE): canonical scalars (< r),
// canonical coordinates (< p), on-curve.  bad[0] counts offenders.
__global__ void k_validate(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ points, uint32_t n,
                           int* __restrict__ bad, uint32_t mont) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = true;
  if (scalars) {
    constexpr uint32_t r[8] = SNARKV_FR_R_LIMBS;
    uint32_t k[8];
    load_words16(scalars + (size_t)i * 8, k, 2);
    uint64_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t x = (uint64_t)k[j] - r[j] - borrow;
      borrow = (x >> 32) & 1u;
    }
    ok = ok && (borrow != 0);
  }
  if (points) {
    uint32_t pw[16];
    load_words16(points + (size_t)i * 16, pw, 4);
    ok = ok && fq_canonical_in_range(pw) && fq_canonical_in_range(pw + 8);
    if (ok) {
      G1Affine q;  // fq.h's domain is R = 2^256: the in-memory form IS its representation
      if (mont) {
#pragma unroll
        for (int j = 0; j < 8; ++j) q.x.v[j] = pw[j], q.y.v[j] = pw[8 + j];
      } else {
        q = g1a_from_canonical(pw);
      }
      ok = g1a_is_on_curve(q);
    }
  }
  if (!ok) atomicAdd(bad, 1);
}

// chunks per half-scalar: as many as keep the chunk kernel at <= ~1 wavefront
// per SIMD (beyond that the extra additions cost more than the shorter chain saves)
static uint32_t chunks_for(size_t n_terms) {
  if (const char* e = getenv("SNARKV_NAIVE_CHUNKS")) {
    int v = atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) return (uint32_t)v;
  }
  if (n_terms <= 2048) return 16;
  if (n_terms <= 4096) return 8;
  if (n_terms <= 8192) return 4;
  if (n_terms <= 16384) return 2;
  return 1;
}

int launch_msm_batched(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, const void* d_offsets,
                       size_t n_msm, size_t n_terms, void* d_out) {
  void* d_terms = nullptr;
  const uint32_t mont = ctx->mont ? 1u : 0u;  // SNARKV_FLAG_MONTGOMERY: terms in and points out in halo2curves' in-memory form
  SNARKV_TRY(ctx_reserve(ctx, SLOT_TERM_PARTIALS, 2 * n_terms * sizeof(G1Xyzz29), &d_terms));
  const uint32_t J = chunks_for(n_terms);
  if (J == 1) {
    // throughput-bound launches (tens of thousands of terms, or other launches in flight next to this context's: its
    // throughput hint) take the grouped form: a third of the issued work on a chain K times as long
    const char* ej = getenv("SNARKV_NAIVE_JOINT");  // 0 two lanes per term / 1 groups (K chosen on the device) / 2, 3, 4: groups of that K (test / A-B knob)
    const int jmode = ej ? atoi(ej) : ((n_terms >= 49152 || ctx->throughput_mode) ? 1 : 0);
    void* d_tab = nullptr;  // the fixed-window tables, limb-major per lane
    if (jmode >= 1) {
      // several terms of a segment per lane on shared doublings; the lane -> segment map is built on the device
      static int slots = 0;  // resident wavefronts of this kernel: 2 per SIMD
      if (!slots) {
        hipDeviceProp_t prop;
        slots = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0 ? 8 * prop.multiProcessorCount : 2048;
      }
      const uint32_t nblk = (uint32_t)((n_msm + 1023) / 1024);
      const size_t max_lanes = n_terms / 2 + n_msm + 1;
      const uint32_t grid = (uint32_t)std::min<size_t>((size_t)slots, (max_lanes + 63) / 64);
      void* d_map = nullptr;
      SNARKV_TRY(ctx_reserve(ctx, SLOT_TERM_MAGS, ((size_t)3 * nblk + n_msm + 1 + 8) * 4, &d_map));
      uint32_t* bsum = (uint32_t*)d_map;
      uint32_t* choice = bsum + 3 * (size_t)nblk;
      uint32_t* base = choice + 8;
      SNARKV_TRY(ctx_reserve(ctx, SLOT_TERM_CHAIN, (size_t)grid * kGroupMax * kGroupRows * 64 * 4, &d_tab));
      hipLaunchKernelGGL(k_gmap_count, dim3(nblk), dim3(256), 0, ctx->stream, (const uint32_t*)d_offsets, (uint32_t)n_msm, bsum);
      // the slots THIS launch can count on: the whole machine when it runs alone; under the throughput hint other launches
      // share it -- as many as the hint / the context pool says (16 when it does not: what bench.py keeps in flight) -- and
      // the choice then goes by total work, as it should when the machine is full whatever this launch does
      const int peers = ctx->throughput_peers > 1 ? ctx->throughput_peers : 16;
      const uint32_t my_slots = ctx->throughput_mode ? (uint32_t)std::max(1, slots / peers) : (uint32_t)slots;
      hipLaunchKernelGGL(k_gmap_choose, dim3(1), dim3(1024), 0, ctx->stream, bsum, nblk, choice, (uint32_t)(jmode >= 2 ? jmode : 0), my_slots);
      hipLaunchKernelGGL(k_gmap_fill, dim3(nblk), dim3(256), 0, ctx->stream, (const uint32_t*)d_offsets, (uint32_t)n_msm,
                         (const uint32_t*)bsum, (const uint32_t*)choice, base);
      hipLaunchKernelGGL(k_term_scalar_mul_group, dim3(grid), dim3(64), 0, ctx->stream, (const uint32_t*)d_scalars,
                         (const uint32_t*)d_points, (G1Xyzz29*)d_terms, (const uint32_t*)d_offsets, (uint32_t)n_msm,
                         (const uint32_t*)base, (const uint32_t*)choice, (int32_t*)d_tab, mont);
    } else {
      uint32_t blocks = (uint32_t)((2 * n_terms + 63) / 64);
      SNARKV_TRY(ctx_reserve(ctx, SLOT_TERM_CHAIN, (size_t)blocks * 3 * 36 * 64 * 4, &d_tab));
      hipLaunchKernelGGL(k_term_scalar_mul, dim3(blocks), dim3(64), 0, ctx->stream, (const uint32_t*)d_scalars,
                         (const uint32_t*)d_points, (G1Xyzz29*)d_terms, (uint32_t)n_terms, (int32_t*)d_tab, mont);
    }
  } else {
    void* d_chain = nullptr;
    void* d_mags = nullptr;
    SNARKV_TRY(ctx_reserve(ctx, SLOT_TERM_CHAIN, 2 * n_terms * J * sizeof(G1Xyzz29), &d_chain));
    SNARKV_TRY(ctx_reserve(ctx, SLOT_TERM_MAGS, 2 * n_terms * sizeof(uint4), &d_mags));
    const uint32_t bits = 128 / J;
    uint32_t blocks_a = (uint32_t)((2 * n_terms + 63) / 64);
    const char* eq = getenv("SNARKV_NAIVE_QUAD");  // 0 / 1 force the one-lane / four-lane chain (test knob)
    const bool quad = eq ? atoi(eq) != 0 : 8 * n_terms <= 262144;  // four lanes per chain while they all fit the wave slots
    if (quad)
      hipLaunchKernelGGL(k_term_chain_quad, dim3((uint32_t)((8 * n_terms + 63) / 64)), dim3(64), 0, ctx->stream,
                         (const uint32_t*)d_scalars, (const uint32_t*)d_points, (G1Xyzz29*)d_chain, (uint4*)d_mags,
                         (uint32_t)n_terms, J, bits, mont);
    else
      hipLaunchKernelGGL(k_term_chain, dim3(blocks_a), dim3(64), 0, ctx->stream, (const uint32_t*)d_scalars,
                         (const uint32_t*)d_points, (G1Xyzz29*)d_chain, (uint4*)d_mags, (uint32_t)n_terms, J, bits, mont);
    uint32_t n_lanes = (uint32_t)(2 * n_terms * J);
    hipLaunchKernelGGL(k_term_chunks, dim3((n_lanes + 63) / 64), dim3(64), 0, ctx->stream,
                       (const G1Xyzz29*)d_chain, (const uint4*)d_mags, (G1Xyzz29*)d_terms, n_lanes, J, bits);
  }
  const uint32_t nm = (uint32_t)n_msm;
  if (n_terms >= 128 * n_msm)
    hipLaunchKernelGGL((k_segment_fold<256, 256>), dim3(nm), dim3(256), 0, ctx->stream, (const G1Xyzz29*)d_terms,
                       (const uint32_t*)d_offsets, (uint32_t*)d_out, nm, mont);
  else if (n_terms > 32 * n_msm)
    hipLaunchKernelGGL((k_segment_fold<64, 64>), dim3(nm), dim3(64), 0, ctx->stream, (const G1Xyzz29*)d_terms,
                       (const uint32_t*)d_offsets, (uint32_t*)d_out, nm, mont);
  else  // <= 64 partials per MSM on average: four MSMs per wavefront
    hipLaunchKernelGGL((k_segment_fold<64, 16>), dim3((nm + 3) / 4), dim3(64), 0, ctx->stream, (const G1Xyzz29*)d_terms,
                       (const uint32_t*)d_offsets, (uint32_t*)d_out, nm, mont);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

int launch_validate(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int* bad_host) {
  void* d_bad = nullptr;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_FLAGS, 64, &d_bad));
  SNARKV_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
  uint32_t blocks = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(k_validate, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t*)d_scalars,
                     (const uint32_t*)d_points, (uint32_t)n, (int*)d_bad, ctx->mont ? 1u : 0u);
  SNARKV_HIP(hipGetLastError());
  SNARKV_HIP(hipMemcpyAsync(bad_host, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  return SNARKV_OK;
}

}  // namespace snarkv
