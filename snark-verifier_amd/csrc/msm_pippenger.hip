// Large BN254 G1 MSM: windowed-bucket Pippenger, re-designed for gfx950.
// Replaces `util::msm::multi_scalar_multiplication`
// (reference snark-verifier/src/util/msm.rs:259-343).
//
// The reference is a serial textbook Pippenger: unsigned c-bit digits,
// c = ceil(ln n)+2, windows walked top-down with c doublings in between, one
// bucket array reused per window (msm.rs:268-302).  On a 256-CU machine that
// shape has no parallelism, so the device algorithm is different while
// producing the same group element (canonical after `to_affine`):
//
//   P0 k_prepare        GLV split k = k1 + k2*lambda (|k_i| < 2^127, glv.h) and
//                        points P, phi(P) = (beta x, y): canonical LE -> 9x29-bit
//                        Montgomery, once.  2n half-width terms: the same number
//                        of bucket additions, half the windows.  Fused with S1:
//                        signed c-bit digits (half the buckets of msm.rs:269),
//                        per-tile LDS histogram of (window, high digit bits)
//   S2 k_scan_*          exclusive scan of the key x tile matrix
//   S3 k_sort_scatter_staged  stable partition of (bucket, point, sign) by
//                        (window, high bits) -- LDS cursors, no global atomics
//   S4 k_sort_level2     one workgroup per (window, high bits): LDS counting
//                        sort by the low digit bits; emits the bucket-sorted
//                        stream plus per-bucket counts/offsets
//   P4 k_accumulate      every lane owns a FIXED-LENGTH run of the sorted
//                        stream (kRun entries) -- load-balanced whatever the
//                        scalar distribution -- and emits a head partial, a
//                        tail partial and complete interior buckets
//                        (the `buckets[d-1].add_assign(base)` of msm.rs:291-296)
//   P5 k_combine         per bucket: stitch the partials of the runs it spans
//   P6 k_bucket_reduce   running-sum trick of msm.rs:298-302, chunked so that
//                        >= 64 K lanes work; the chunk weights (chunk index x
//                        chunk sum) come from one suffix scan over the 64 lanes
//                        of a block instead of a per-lane double-and-add; a
//                        block leaves (weighted sum T, plain sum S)
//   P8 k_shift_windows   one wavefront per window: the same suffix-scan fold
//                        one level up (block index x S), then T_w = 2^(c w) S_w
//                        (the `result.double()` x c of msm.rs:285-287) -- the
//                        dependency chain GLV halves -- with the 7 products of
//                        a Jacobian doubling spread over 3 lanes (depth 3)
//   P9 k_final           sum of the shifted window sums + `to_affine`, or the
//                        projective partial for the multi-GPU fold.
//
// Field arithmetic is the lazy 9x29-bit form (fq29.h, g1_29.h): branch-free
// "fast" adders, one degenerate-ZZ check per run and a careful redo only when an
// exceptional case (P = +-Q, identity) was met.
//
// MFMA is deliberately unused: there is no dense contraction, the work is
// 254-bit modular multiplication on the integer VALU (v_mad_i64_i32).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <atomic>
#include "ctx.hpp"
#include "g1_29.h"
#include "fr29.h"
#ifndef SNARKV_GLV
#define SNARKV_GLV 1  // 0: curves without the BN-shaped GLV lattice (the pasta build): one virtual point per point
#endif
#if SNARKV_GLV
#include "glv.h"
#endif

namespace snarkv {

#ifndef SNARKV_KRUN
#define SNARKV_KRUN 64
#endif
#ifndef SNARKV_KCHUNK
#define SNARKV_KCHUNK 8
#endif
#ifndef SNARKV_TILE_THREADS
// Tile workgroups of k_prepare / k_sort_scatter_staged: 256 lanes on 2 048 scalars, i.e. ONE wavefront per SIMD, and registers
// capped so that one fits beside three resident k_accumulate wavefronts (3 x 136 VGPRs leave 104 per SIMD): with several
// MSMs in flight the sorts of the next MSM then run UNDER the accumulation instead of waiting for its wavefronts to
// retire.  Measured against 512 lanes on 4 096 scalars (two wavefronts per SIMD, 120 VGPRs): batch of 40 MSMs -1.5 to
// -3.4 %, one MSM alone level (profiles/r02_ab_coresidency.txt).
#define SNARKV_TILE_THREADS 256
#endif
#ifndef SNARKV_SCATTER_ATTR
#define SNARKV_SCATTER_ATTR __attribute__((amdgpu_waves_per_eu(5)))  // k_sort_scatter_staged: 96 VGPRs (120 uncapped, no spills either way)
#endif
#ifndef SNARKV_L2_THREADS
#define SNARKV_L2_THREADS 512  // lanes of a k_sort_level2 workgroup
#endif
#ifndef SNARKV_LEVEL2_ATTR
#define SNARKV_LEVEL2_ATTR
#endif
#ifndef SNARKV_PREP_THREADS
#define SNARKV_PREP_THREADS SNARKV_TILE_THREADS  // lanes of a k_prepare workgroup (one tile)
#endif
#ifndef SNARKV_XCD_TILES
#define SNARKV_XCD_TILES 1  // tiles -> workgroups so that an XCD owns a contiguous tile range (xcd_tile)
#endif
#ifndef SNARKV_TILE_BASE
#define SNARKV_TILE_BASE 2048  // smallest tile (scalars per k_prepare / k_sort_scatter_staged workgroup)
#endif
#ifndef SNARKV_ACC_WAVES
#define SNARKV_ACC_WAVES 3
#endif
// Wave priority of every Pippenger kernel EXCEPT k_accumulate (s_setprio, 0..3).  With several MSMs in flight the
// latency-bound stages (one wavefront per SIMD or per window: bucket reduce, shift chains, to_affine) share their
// SIMDs with three resident k_accumulate wavefronts of a neighbouring MSM; at equal priority the issue arbiter
// gives them a quarter of the slots and they run ~3x longer than alone (rocprofv3 trace: k_shift_windows
// 0.35 -> 0.57-0.71 ms, k_sort_level2 0.08 -> 0.8 ms), which is what the in-flight plateau is made of.  At
// priority 3 they issue whenever they are ready -- a dependency chain cannot use more than its own latency allows --
// and k_accumulate fills every other slot.
#ifndef SNARKV_BPRIO
#define SNARKV_BPRIO 3
#endif
#define SNARKV_RAISE_PRIO() __builtin_amdgcn_s_setprio(SNARKV_BPRIO)

constexpr int kHalves = SNARKV_GLV ? 2 : 1;       // virtual points per input point: P and phi(P), or P alone
constexpr int kDigitWords = SNARKV_GLV ? 4 : 8;   // words of a digit source: a 127-bit GLV half / the 255-bit scalar
constexpr int kDigitBits = 32 * kDigitWords;      // W * c covers this: magnitude bits + the recoding carry
constexpr int kRun = SNARKV_KRUN;      // P4: entries per lane (a context with the throughput hint uses kRunThroughput)
#ifndef SNARKV_KRUN_THROUGHPUT
#define SNARKV_KRUN_THROUGHPUT 96
#endif
// Longer runs = fewer head/tail partials for k_combine and fewer bucket-head entries that occupy an addition slot without
// adding (interleaved A/B, profiles/r02_ab_krun*.txt: 64 -> 96 entries = -3.5 % per MSM with several MSMs in flight),
// but 2^24 entries / 96 / 64 = 2 731 wavefronts no longer fill the 3 072 slots of the machine: one MSM alone runs
// k_accumulate 1.11 -> 1.26 ms.  So the run length follows the context's hint (snarkv_ctx_set_throughput_hint).
constexpr int kRunThroughput = SNARKV_KRUN_THROUGHPUT;
constexpr int kChunk = SNARKV_KCHUNK;  // P6: buckets per lane
constexpr uint32_t kSortCap = 7168;     // S4: items a workgroup sorts entirely in LDS (56 KiB of the 64 KiB dynamic limit)
constexpr uint32_t kSortTarget = 3072;  // S1: average items per (window, high bits) key
constexpr uint32_t kMaxKeys = 16384;    // S1: LDS counters per tile workgroup (64 KiB)
constexpr uint32_t kBigSpan = 24;       // P5: buckets spanning more runs go to the cooperative kernel
constexpr uint32_t kBigGrid = 8192;  // P5: workgroups of k_combine_big (they walk the list of oversized buckets; 128 measured level)
constexpr uint32_t kMaxBig = 8192;   // S1: (window, high bits) keys per window <= 128
constexpr uint32_t kNoBucket = 0xFFFFFFFFu;

struct PipParams {
  uint32_t n;
  int c;           // window bits
  int W;           // windows = ceil(128 / c)  (GLV half-scalars, |k| < 2^127)
  uint32_t B;      // buckets per window = 2^(c-1)  (signed digits)
  uint32_t nb;     // W * B
  int low_bits;    // S4 sorts by these low bits of (|digit|-1)
  uint32_t SB;     // keys per window at level 1 = B >> low_bits
  uint32_t nkeys;  // W * SB
  uint32_t nblk;   // tiles
  uint32_t tile;   // scalars per tile workgroup
  uint32_t mstride;  // row stride of the key x tile matrix (odd: no power-of-two channel aliasing)
  uint32_t w0;       // index of the first window held (bucket-sharded reduce of a window range; 0 otherwise)
  uint32_t krun;     // entries per run (kRun, or kRunThroughput on a context with the throughput hint)
  uint32_t wper;     // batched tail over several MSMs' grids laid end to end: windows per MSM (0: one MSM)
  uint32_t chunk_log2;  // P6: log2 of the buckets per k_bucket_reduce lane (kLog2Chunk; larger in a batched tail)
  uint32_t mont;     // P0: 1 = scalars and points arrive in halo2curves' in-memory form (a * 2^256 mod r / mod p): SNARKV_FLAG_MONTGOMERY
};

// c bits at offset lo of a kDigitBits-bit magnitude held in registers (selects, no dynamic indexing)
__device__ __forceinline__ uint32_t half_bits(const uint32_t (&k)[kDigitWords], int lo, int c) {
  if (lo >= kDigitBits) return 0;
  int word = lo >> 5, sh = lo & 31;
  uint32_t w0 = 0u, w1 = 0u;
#pragma unroll
  for (int i = 0; i < kDigitWords; ++i) {
    w0 = word == i ? k[i] : w0;
    if (i > 0) w1 = word == i - 1 ? k[i] : w1;
  }
  uint64_t v = ((uint64_t)w1 << 32) | w0;
  return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// --------------------------------------------------------------- P0 + S1
// Signed-digit recoding of a 127-bit magnitude: raw = bits + carry in [0, 2^c];
// raw > 2^(c-1) becomes raw - 2^c (negative) with a carry into the next window.
// With W*c >= 128 the top window never carries out.  A negative half (sign in
// bit 127) flips every digit's sign.  `emit(key, bucket, neg)` per non-zero digit.
template <class F>
__device__ __forceinline__ void for_each_digit(const uint32_t (&k)[kDigitWords], uint32_t sgn, const PipParams& p, F emit) {
  uint32_t carry = 0;
  for (int w = 0; w < p.W; ++w) {
    uint32_t raw = half_bits(k, w * p.c, p.c) + carry;
    uint32_t neg = raw > p.B ? 1u : 0u;
    uint32_t d = neg ? ((1u << p.c) - raw) : raw;
    carry = neg;
    if (d != 0) emit((uint32_t)w * p.SB + ((d - 1) >> p.low_bits), (uint32_t)w * p.B + d - 1, neg ^ sgn);
  }
}
// One window's digit of the same recoding, without walking the windows below it: the carry into window w is 1 iff the
// low w*c bits exceed the constant with B in every window, i.e. the first window below w (from the top) whose bits
// differ from B decides (expected: one step).
template <class F>
__device__ __forceinline__ void digit_of_window(const uint32_t (&k)[kDigitWords], uint32_t sgn, const PipParams& p, int w,
                                                F emit) {
  uint32_t carry = 0;
  for (int j = w - 1; j >= 0; --j) {
    uint32_t b = half_bits(k, j * p.c, p.c);
    if (b != p.B) {
      carry = b > p.B ? 1u : 0u;
      break;
    }
  }
  uint32_t raw = half_bits(k, w * p.c, p.c) + carry;
  uint32_t neg = raw > p.B ? 1u : 0u;
  uint32_t d = neg ? ((1u << p.c) - raw) : raw;
  if (d != 0) emit((uint32_t)w * p.SB + ((d - 1) >> p.low_bits), (uint32_t)w * p.B + d - 1, neg ^ sgn);
}
// digit source held in registers (k_prepare): copy + sign split
__device__ __forceinline__ uint32_t load_digit_words(const uint32_t* src, uint32_t (&k)[kDigitWords]) {
#pragma unroll
  for (int j = 0; j < kDigitWords; ++j) k[j] = src[j];
#if SNARKV_GLV
  uint32_t sgn = k[3] >> 31;
  k[3] &= 0x7FFFFFFFu;
  return sgn;
#else
  return 0u;
#endif
}
// the stored digit source of virtual point v: kDigitWords words; with GLV bit 127 is the half's sign
__device__ __forceinline__ uint32_t load_digit_source(const uint4* __restrict__ src, size_t v, uint32_t (&k)[kDigitWords]) {
  const uint4* s = src + v * (kDigitWords / 4);
#pragma unroll
  for (int j = 0; j < kDigitWords / 4; ++j) {
    uint4 q = s[j];
    k[4 * j] = q.x;
    k[4 * j + 1] = q.y;
    k[4 * j + 2] = q.z;
    k[4 * j + 3] = q.w;
  }
#if SNARKV_GLV
  uint32_t sgn = k[3] >> 31;
  k[3] &= 0x7FFFFFFFu;
  return sgn;
#else
  return 0u;  // canonical scalars < r < 2^255: non-negative, bit 255 clear
#endif
}

// Workgroup b runs on XCD b % 8 (observed placement, MI355X_MICROARCH.md): give every XCD a CONTIGUOUS range of tiles, so
// that the sorted stream's lines shared by neighbouring tiles' runs of one key collect both halves in ONE L2.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t nblk) {
#if SNARKV_XCD_TILES
  return (nblk & 7u) == 0 ? (b & 7u) * (nblk >> 3) + (b >> 3) : b;
#else
  return b;
#endif
}

// One workgroup = one tile of p.tile scalars: GLV split k = k1 + k2*lambda,
// P and phi(P) = (beta x, y) to 9x29-bit Montgomery, and the LDS histogram of
// the (window, high digit bits) keys -> column `blockIdx` of the matrix M.
// The identity (64 zero bytes) contributes nothing: its half-scalars are
// stored as zero, so the sort never has to look at the points again.
__global__ void __launch_bounds__(SNARKV_PREP_THREADS)
    k_prepare(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ points,
              G1Packed* __restrict__ pts, uint4* __restrict__ glv, PipParams p, uint32_t* __restrict__ M) {
  SNARKV_RAISE_PRIO();
  extern __shared__ uint32_t lds[];  // nkeys counters
  for (uint32_t k = threadIdx.x; k < p.nkeys; k += blockDim.x) lds[k] = 0u;
  __syncthreads();
  const uint32_t tile = xcd_tile(blockIdx.x, p.nblk);
  uint32_t lo = tile * p.tile;
  uint32_t hi = lo + p.tile < p.n ? lo + p.tile : p.n;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const uint4* s = reinterpret_cast<const uint4*>(points + (size_t)i * 16);
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 v = s[j];
      w[4 * j] = v.x;
      w[4 * j + 1] = v.y;
      w[4 * j + 2] = v.z;
      w[4 * j + 3] = v.w;
    }
    G1Affine29 a = g1a29_from_words(w, p.mont != 0);  // one product by a constant either way
    bool ident = g1a29_is_identity(a);
    pts[kHalves * (size_t)i] = g1a29_pack(a);
    const uint4* ks = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    uint4 k0 = ks[0], k1 = ks[1];
    uint32_t k[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w}, o[8];
    if (p.mont) {  // the scalar's in-memory form -> its canonical integer: one Fr product more (~12 % of this kernel)
      uint32_t kc[8];
      fr_words_from_mont256(k, kc);
#pragma unroll
      for (int j = 0; j < 8; ++j) k[j] = kc[j];
    }
#if SNARKV_GLV
    constexpr int32_t bl[9] = SNARKV_GLV_BETA29_LIMBS;
    Fq29 beta;
#pragma unroll
    for (int j = 0; j < 9; ++j) beta.v[j] = bl[j];
    a.x = fq29_canon_of_product(fq29_mul(a.x, beta));  // phi(P) = (beta x, y)
    pts[2 * (size_t)i + 1] = g1a29_pack(a);
    glv_decompose(k, o);
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = k[j];  // the canonical scalar is its own (only) digit source
#endif
    if (ident) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0;
    }
    glv[2 * (size_t)i] = make_uint4(o[0], o[1], o[2], o[3]);
    glv[2 * (size_t)i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
#pragma unroll
    for (int h = 0; h < kHalves; ++h) {
      uint32_t d[kDigitWords];
      uint32_t sgn = load_digit_words(o + h * kDigitWords, d);
      for_each_digit(d, sgn, p, [&](uint32_t key, uint32_t, uint32_t) { atomicAdd(&lds[key], 1u); });
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < p.nkeys; k += blockDim.x) M[(size_t)k * p.mstride + tile] = lds[k];
  // the padding column of the matrix (mstride = nblk | 1) must read as zero in the scan: one workgroup writes it
  // (the per-MSM fills are done by the kernels that own the data, not by hipMemsetAsync: profiles/r03_ab_scheduling.txt)
  if (blockIdx.x == 0 && p.mstride > p.nblk)
    for (uint32_t k = threadIdx.x; k < p.nkeys; k += blockDim.x) M[(size_t)k * p.mstride + p.nblk] = 0u;
}

// --------------------------------------------------------------- S3
// Stable partition by (window, high bits).  M has been scanned (key-major, tile-minor), so M[key][tile] is where this
// tile's items of that key start.  The writes are STAGED through LDS: a workgroup owns ONE tile of kStageScalars scalars
// and per window counts its entries per key, scans the counts, places the entries in LDS in key order and writes them
// out with consecutive lanes on consecutive addresses: a (tile, key) run leaves as whole 64-128-byte segments instead
// of 8-byte stores issued at unrelated times (the L2 does not merge those: WRITE_SIZE was 3.5x the payload with direct
// stores, 1.0x staged -- profiles/r02_pmc_hbm_traffic.txt; the direct kernel: git tag exp/scatter-direct).
constexpr uint32_t kStageScalars = SNARKV_TILE_BASE;                    // scalars per staged workgroup = the tile
constexpr uint32_t kStageItems = kStageScalars / SNARKV_TILE_THREADS;   // ... per lane
static_assert(kStageScalars % SNARKV_TILE_THREADS == 0, "a staged tile gives every lane the same number of scalars");
__global__ void __launch_bounds__(SNARKV_TILE_THREADS) SNARKV_SCATTER_ATTR
    k_sort_scatter_staged(const uint4* __restrict__ glv, PipParams p, const uint32_t* __restrict__ M,
                          uint2* __restrict__ tmp) {
  SNARKV_RAISE_PRIO();
  extern __shared__ uint32_t lds[];  // gbase[SB] | cnt[SB] | off[SB] | wsum[16] | (pad) | stage[kStageScalars * kHalves] (uint2)
  uint32_t* gbase = lds;
  uint32_t* cnt = lds + p.SB;
  uint32_t* off = lds + 2 * p.SB;
  uint32_t* wsum = lds + 3 * p.SB;
  uint2* stage = reinterpret_cast<uint2*>(lds + 3 * p.SB + 16 + (p.SB & 1u));  // 8-byte aligned (SB = 1 for tiny MSMs)
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t tile = xcd_tile(blockIdx.x, p.nblk);
  const uint32_t lo = tile * kStageScalars;
  const uint32_t hi = lo + kStageScalars < p.n ? lo + kStageScalars : p.n;
  const uint32_t per = (p.SB + SNARKV_TILE_THREADS - 1) / SNARKV_TILE_THREADS;  // keys a lane scans
  // (Keeping the lane's 16 digit sources in registers across the windows instead of re-reading them -- FETCH 270 -> 40 MB
  // -- made the kernel 12 % faster alone and the batch 5 % SLOWER: 196 VGPRs, and a 512-lane workgroup of those finds no
  // room next to a resident k_accumulate.  The re-reads come out of L2 / the Infinity Cache.)
  for (int w = 0; w < p.W; ++w) {
    for (uint32_t k = tid; k < p.SB; k += blockDim.x) {
      gbase[k] = M[((size_t)w * p.SB + k) * p.mstride + tile];
      cnt[k] = 0u;
    }
    __syncthreads();
    // this lane's entries of window w: bucket inside the window (0xFFFFFFFF: none), rank inside its key, sign
    uint32_t ebkt[kStageItems * kHalves], erank[kStageItems * kHalves], eneg = 0u;
#pragma unroll
    for (uint32_t j = 0; j < kStageItems; ++j) {
      const uint32_t i = lo + tid + j * SNARKV_TILE_THREADS;
#pragma unroll
      for (uint32_t h = 0; h < (uint32_t)kHalves; ++h) {
        const uint32_t e = j * kHalves + h;
        ebkt[e] = 0xFFFFFFFFu;
        if (i < hi) {
          uint32_t d[kDigitWords];
          uint32_t sgn = load_digit_source(glv, (size_t)kHalves * i + h, d);
          digit_of_window(d, sgn, p, w, [&](uint32_t key, uint32_t bucket, uint32_t neg) {
            ebkt[e] = bucket - (uint32_t)w * p.B;
            eneg |= neg << e;
            erank[e] = atomicAdd(&cnt[key - (uint32_t)w * p.SB], 1u);
          });
        }
      }
    }
    __syncthreads();
    // exclusive scan of cnt[0 .. SB) -> off
    uint32_t mine = 0;
    for (uint32_t k = tid * per; k < (tid + 1) * per && k < p.SB; ++k) mine += cnt[k];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t up = __shfl_up(incl, d, 64);
      if ((int)lane >= d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t q = 0; q < wave; ++q) run += wsum[q];
    for (uint32_t k = tid * per; k < (tid + 1) * per && k < p.SB; ++k) {
      off[k] = run;
      run += cnt[k];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < kStageItems * kHalves; ++e)
      if (ebkt[e] != 0xFFFFFFFFu) {
        const uint32_t v = kHalves * (lo + tid + (e / kHalves) * SNARKV_TILE_THREADS) + (e % kHalves);
        stage[off[ebkt[e] >> p.low_bits] + erank[e]] = make_uint2(ebkt[e] + (uint32_t)w * p.B, v | (((eneg >> e) & 1u) << 31));
      }
    __syncthreads();
    const uint32_t total = off[p.SB - 1] + cnt[p.SB - 1];
    for (uint32_t idx = tid; idx < total; idx += blockDim.x) {
      const uint2 e = stage[idx];
      const uint32_t key = (e.x - (uint32_t)w * p.B) >> p.low_bits;
      tmp[gbase[key] + (idx - off[key])] = e;
    }
    __syncthreads();  // the next window overwrites gbase / cnt / stage
  }
}

// --------------------------------------------------------------- S2
// In-place exclusive scan of `nb` counters: 1024 per block (256 lanes x 4),
// block sums scanned by one block, then added back.
__global__ void __launch_bounds__(256) k_scan_local(uint32_t* __restrict__ data, uint32_t* __restrict__ blocksum,
                                                    uint32_t nb) {
  SNARKV_RAISE_PRIO();
  __shared__ uint32_t sh[256];
  uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
  uint32_t v[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = (base + j < nb) ? data[base + j] : 0;
    s += v[j];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t off = 1; off < 256; off <<= 1) {
    uint32_t t = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t excl = sh[threadIdx.x] - s;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < nb) data[base + j] = excl;
    excl += v[j];
  }
  if (threadIdx.x == 255) blocksum[blockIdx.x] = sh[255];
}

__global__ void __launch_bounds__(1024) k_scan_blocksums(uint32_t* __restrict__ blocksum, uint32_t nblocks,
                                                         uint32_t* __restrict__ total_out) {
  SNARKV_RAISE_PRIO();
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t running;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblocks; base += 1024) {
    uint32_t idx = base + threadIdx.x;
    uint32_t v = idx < nblocks ? blocksum[idx] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      uint32_t t = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    uint32_t r0 = running;
    if (idx < nblocks) blocksum[idx] = r0 + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) running = r0 + sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = running;
}

__global__ void __launch_bounds__(256) k_scan_add(uint32_t* __restrict__ data, const uint32_t* __restrict__ blocksum,
                                                  uint32_t nb) {
  SNARKV_RAISE_PRIO();
  uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
  uint32_t add = blocksum[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < nb) data[base + j] += add;
}

// --------------------------------------------------------------- S4
// One workgroup per level-1 key (window, high bits): counting sort of its
// slice of `tmp` by the low digit bits with LDS counters.  Slices of up to
// kSortCap items (the normal case: S1 sizes the keys for it) are sorted
// ENTIRELY in LDS and written back as one coalesced stream -- a random 8-byte
// scatter to HBM costs a 128-byte read-modify-write once the working set
// outgrows L2/Infinity Cache.  Larger slices (skewed scalars) take the
// two-pass global path.
__global__ void __launch_bounds__(SNARKV_L2_THREADS) SNARKV_LEVEL2_ATTR
    k_sort_level2(const uint2* __restrict__ tmp, const uint32_t* __restrict__ M, uint32_t* __restrict__ misc,
                  PipParams p, uint2* __restrict__ entries, uint32_t* __restrict__ counts,
                  uint32_t* __restrict__ offsets) {
  SNARKV_RAISE_PRIO();
  extern __shared__ uint32_t lds[];  // nbins counters | one scan word per lane | kSortCap items (uint2)
  const uint32_t nbins = 1u << p.low_bits;
  uint32_t* hist = lds;
  uint32_t* scan = lds + nbins;
  uint2* stage = reinterpret_cast<uint2*>(lds + nbins + SNARKV_L2_THREADS + (nbins & 1u));  // 8-byte aligned (one bin for tiny windows)
  const uint32_t T = blockDim.x;
  uint32_t key = blockIdx.x;
  uint32_t begin = M[(size_t)key * p.mstride];
  uint32_t end = (key + 1 < p.nkeys) ? M[(size_t)(key + 1) * p.mstride] : misc[0];
  uint32_t cnt_items = end - begin;
  bool fast = cnt_items <= kSortCap;
  for (uint32_t k = threadIdx.x; k < nbins; k += T) hist[k] = 0;
  __syncthreads();
  const uint32_t low_mask = nbins - 1;
  constexpr int kPer = kSortCap / SNARKV_L2_THREADS;  // items a lane keeps in registers on the fast path
  static_assert(kSortCap % SNARKV_L2_THREADS == 0, "kSortCap items spread evenly over the lanes");
  uint2 mine[kPer];
  if (fast) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      uint32_t e = threadIdx.x + j * T;
      if (e < cnt_items) {
        mine[j] = tmp[begin + e];
        atomicAdd(&hist[mine[j].x & low_mask], 1u);
      }
    }
  } else {
    for (uint32_t e = begin + threadIdx.x; e < end; e += T) atomicAdd(&hist[tmp[e].x & low_mask], 1u);
  }
  __syncthreads();
  // exclusive scan of the bin sizes: each lane owns a contiguous strip
  uint32_t per = (nbins + T - 1) / T;
  uint32_t s0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t k = s0; k < s0 + per && k < nbins; ++k) sum += hist[k];
  scan[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < T; off <<= 1) {
    uint32_t t = (threadIdx.x >= off) ? scan[threadIdx.x - off] : 0;
    __syncthreads();
    scan[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = scan[threadIdx.x] - sum;  // position inside the key's slice
  uint32_t w = key / p.SB, sb = key % p.SB;
  uint32_t bucket0 = w * p.B + (sb << p.low_bits);
  for (uint32_t k = s0; k < s0 + per && k < nbins; ++k) {
    uint32_t cnt = hist[k];
    counts[bucket0 + k] = cnt;
    offsets[bucket0 + k] = begin + run;
    hist[k] = run;  // becomes the cursor (slice-relative)
    run += cnt;
  }
  // the big-bucket counter of k_combine (a per-MSM fill done by the kernel that runs before its user, not a memset)
  if (key == p.nkeys - 1 && threadIdx.x == 0) misc[4] = 0u;
  __syncthreads();
  if (fast) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      uint32_t e = threadIdx.x + j * T;
      if (e < cnt_items) stage[atomicAdd(&hist[mine[j].x & low_mask], 1u)] = mine[j];
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < cnt_items; e += T) entries[begin + e] = stage[e];
  } else {
    for (uint32_t e = begin + threadIdx.x; e < end; e += T) {
      uint2 it = tmp[e];
      entries[begin + atomicAdd(&hist[it.x & low_mask], 1u)] = it;
    }
  }
}

// --------------------------------------------------------------- P4
// One lane = one run of kRun consecutive sorted entries.  Emits the partial of
// the run's first bucket (head), of its last bucket if different (tail), and
// writes complete interior buckets straight to `buckets`.  Branch-free adders
// and NO degeneracy test here: a flush is a plain store, so the lanes of a wave
// (which change bucket at different iterations) never wait for each other's
// checks.  P5 tests every bucket once and redoes the rare bad one carefully.
// An entry is {bucket id, point index | sign << 31}; the point is gathered from the Montgomery table (G1Packed: ONE
// 64-byte sector) and its 9 x 29-bit limbs are cut out of the 256-bit words in registers.
// (Measured and removed, round 3: a batched-affine pair level in front of this kernel, in two forms -- bit-exact, 25-37 %
// slower per MSM because its passes are random 64-byte gathers at 3.6-3.8 TB/s: profiles/r03_ab_pair_tree.txt, git tag
// exp/pair-tree.)
constexpr uint32_t kEntryIdx = 0x7FFFFFFFu;  // point index bits of an entry's .y (bit 31 = negate)

__device__ __forceinline__ G1Affine29 entry_point(const G1Packed& k, uint32_t y) {
  G1Affine29 p = g1a29_unpack(k);
  if (y >> 31) p.y = fq29_neg(p.y);
  return p;
}

template <int RUN>
__global__ void __launch_bounds__(64, SNARKV_ACC_WAVES)
    k_accumulate(const uint2* __restrict__ entries, const uint32_t* __restrict__ total_ptr,
                 const G1Packed* __restrict__ pts, G1Xyzz29* __restrict__ buckets, uint32_t* __restrict__ seg_ids,
                 G1Xyzz29* __restrict__ seg_parts) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t stop = *total_ptr;
  const uint64_t begin64 = (uint64_t)t * RUN;
  if (begin64 >= stop) return;
  const uint32_t begin = (uint32_t)begin64;
  const uint32_t end = (stop - begin > (uint32_t)RUN) ? begin + RUN : stop;
  const size_t slot = t;
  uint32_t cur = entries[begin].x;
  bool first = true, fresh = true;
  G1Xyzz29 acc = xyzz29_identity();
  // a bucket's part of the run is finished: head partial, or complete interior bucket
  auto flush = [&]() {
    if (first) {
      seg_ids[2 * slot] = cur;
      seg_parts[2 * slot] = acc;
      first = false;
    } else {
      buckets[cur] = acc;  // complete interior bucket
    }
  };
  // software pipeline: the (entry -> point) gather of step e+1 is issued before the ~2 200-instruction
  // mixed addition of step e; two steps per trip with ping-pong registers, so the prefetched point is
  // consumed where it was loaded instead of being copied (18 moves per entry)
  auto step = [&](const uint2& ent, const G1Packed& pk) {
    if (ent.x != cur) {
      flush();
      cur = ent.x;
      fresh = true;
    }
    G1Affine29 p = entry_point(pk, ent.y);
    if (fresh) {
      acc.x = p.x;
      acc.y = p.y;
      acc.zz = fq29_one();
      acc.zzz = fq29_one();
      fresh = false;
    } else {
      xyzz29_madd_fast(acc, p);
    }
  };
  uint2 ent0 = entries[begin], ent1 = ent0;
  G1Packed p0 = pts[ent0.y & kEntryIdx], p1 = p0;
#pragma unroll 1
  for (uint32_t e = begin; e < end; e += 2) {
    if (e + 1 < end) {
      ent1 = entries[e + 1];
      p1 = pts[ent1.y & kEntryIdx];
    }
    step(ent0, p0);
    if (e + 1 < end) {
      if (e + 2 < end) {
        ent0 = entries[e + 2];
        p0 = pts[ent0.y & kEntryIdx];
      }
      step(ent1, p1);
    }
  }
  if (first) {
    seg_ids[2 * slot] = cur;
    seg_parts[2 * slot] = acc;
    seg_ids[2 * slot + 1] = kNoBucket;
  } else {
    seg_ids[2 * slot + 1] = cur;
    seg_parts[2 * slot + 1] = acc;
  }
}

// --------------------------------------------------------------- P5
// Careful recomputation of one bucket straight from its sorted entries (the
// rare bucket in which a fast addition met P = +-Q: duplicate / opposite bases).
// The result goes to memory (`out`: the bucket itself, or the caller's LDS slot), not back by value: a 144-byte return of
// a non-inlined function travels through the stack (160 bytes of scratch per lane in k_combine until round 4).
__device__ __noinline__ void bucket_from_entries_careful(const uint2* __restrict__ entries, const G1Packed* __restrict__ pts,
                                                         uint32_t o, uint32_t cnt, uint32_t first, uint32_t stride,
                                                         bool sanitize, G1Xyzz29* out) {
  G1Xyzz29 acc = xyzz29_identity();
  for (uint32_t e = o + first; e < o + cnt; e += stride) {
    uint2 ent = entries[e];
    G1Affine29 p = entry_point(pts[ent.y & kEntryIdx], ent.y);
    xyzz29_madd_careful(acc, p);
  }
  *out = sanitize ? xyzz29_sanitize(acc) : acc;
}

// the run slots [s0, s1] that hold partials of a bucket whose entries are entries[o, o + cnt): runs are cut every
// p.krun entries from the start of the sorted stream
__device__ __forceinline__ void run_span(const PipParams& p, uint32_t o, uint32_t cnt, size_t& s0, size_t& s1) {
  s0 = o / p.krun;
  s1 = (o + cnt - 1) / p.krun;
}

// One lane per bucket: stitch the partials of the runs it spans (or pick up the
// value P4 wrote for an interior bucket), then the ONE degeneracy test of the
// bucket: a fast addition that met an exceptional case left ZZ = 0 (mod p)
// (sticky through every later addition; an exact-zero ZZ partial is the same
// signal, a run never legitimately produces the identity).
// (Measured and removed, round 3: run-boundary lanes for the buckets that span exactly two runs -- level in a batch,
// 13 % slower alone: profiles/r03_ab_combine_pack.txt, git tag exp/combine-pairs.)
__global__ void __launch_bounds__(64)
    k_combine(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets, PipParams p,
              const uint2* __restrict__ entries, const G1Packed* __restrict__ pts,
              const uint32_t* __restrict__ seg_ids, const G1Xyzz29* __restrict__ seg_parts,
              G1Xyzz29* __restrict__ buckets, uint32_t* __restrict__ big_count, uint32_t* __restrict__ big_list) {
  SNARKV_RAISE_PRIO();
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.nb) return;
  uint32_t cnt = counts[b];
  if (cnt == 0) {  // empty bucket = the identity (all-zero ZZ): stored here, the grid is never memset
    buckets[b] = xyzz29_identity();
    return;
  }
  uint32_t o = offsets[b];
  size_t s0, s1;
  run_span(p, o, cnt, s0, s1);
  if (s1 - s0 >= kBigSpan) {  // skewed scalars: hand the bucket to k_combine_big
    uint32_t slot = atomicAdd(big_count, 1u);
    if (slot < kMaxBig) {
      big_list[slot] = b;
      return;
    }
  }
  G1Xyzz29 acc = xyzz29_identity();
  bool touched = false, bad = false;
  for (size_t s = s0; s <= s1; ++s) {
    for (int h = 0; h < 2; ++h) {
      if (seg_ids[2 * s + h] == b) {
        G1Xyzz29 part = seg_parts[2 * s + h];
        bad = bad || xyzz29_is_identity(part);
        xyzz29_add_skipid_fast(acc, part, bad);
        touched = true;
      }
    }
  }
  if (!touched) acc = buckets[b];  // interior to one run: P4 stored it
  bad = bad || xyzz29_is_degenerate(acc);
  if (bad) bucket_from_entries_careful(entries, pts, o, cnt, 0, 1, true, &buckets[b]);
  else if (touched) buckets[b] = acc;
}

// Buckets that span many runs (skewed scalar distributions: e.g. all scalars
// equal puts n entries into one bucket per window): one 256-lane workgroup per
// bucket, lane-strided careful adds + LDS tree.  If any partial is degenerate
// the whole bucket is recomputed carefully from its entries.
__global__ void __launch_bounds__(256)
    k_combine_big(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
                  const uint2* __restrict__ entries, const G1Packed* __restrict__ pts,
                  const uint32_t* __restrict__ seg_ids, const G1Xyzz29* __restrict__ seg_parts,
                  G1Xyzz29* __restrict__ buckets, const uint32_t* __restrict__ big_count,
                  const uint32_t* __restrict__ big_list, PipParams p) {
  SNARKV_RAISE_PRIO();
  __shared__ G1Xyzz29 sh[256];
  __shared__ int any_bad;
  uint32_t nbig = *big_count < kMaxBig ? *big_count : kMaxBig;
  // the grid walks the list (normally empty: uniform scalars have no bucket over kBigSpan runs); capping the grid at
  // 128 workgroups was measured level (one MSM alone -0.5 %, the batch +0.9 %)
  for (uint32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    uint32_t b = big_list[bi];
    uint32_t o = offsets[b], cnt = counts[b];
    size_t s0, s1;
    run_span(p, o, cnt, s0, s1);
    if (threadIdx.x == 0) any_bad = 0;
    __syncthreads();
    G1Xyzz29 acc = xyzz29_identity();
    bool bad = false;
    for (size_t s = s0 + threadIdx.x; s <= s1; s += 256)
      for (int h = 0; h < 2; ++h)
        if (seg_ids[2 * s + h] == b) {
          G1Xyzz29 part = seg_parts[2 * s + h];
          bad = bad || xyzz29_is_degenerate(part);
          xyzz29_add_careful(acc, part);
        }
    if (bad) atomicOr(&any_bad, 1);
    __syncthreads();
    if (any_bad) bucket_from_entries_careful(entries, pts, o, cnt, threadIdx.x, 256, false, &sh[threadIdx.x]);
    else sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t st = 128; st >= 1; st >>= 1) {
      if (threadIdx.x < st) {
        G1Xyzz29 a = sh[threadIdx.x];
        xyzz29_add_careful(a, sh[threadIdx.x + st]);
        sh[threadIdx.x] = a;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) buckets[b] = xyzz29_sanitize(sh[0]);
    __syncthreads();  // sh / any_bad are reused by the next bucket of this workgroup
  }
}

// --------------------------------------------------------------- P6
// Lane (w, j) folds buckets [j*kChunk, (j+1)*kChunk) of window w with the
// running-sum trick of msm.rs:298-302:
//   run = sum B_i ;  acc = sum (i - base + 1) B_i ,  base = j*kChunk
// (bucket i has weight i+1).  The lane's missing term base*run is NOT computed
// by a per-lane double-and-add (15 doublings + divergent adds, 60 % of the old
// kernel): lanes of a 64-lane block hold consecutive chunks, so
//   sum_l (L l) run_l = L * sum_{l>=1} Sfx_l ,  Sfx_l = sum_{k>=l} run_k
// -- one suffix scan (6 steps) instead; the block emits T = sum of weights
// relative to ITS first bucket and S = its plain sum, and P8 applies the same
// identity one level up (block index weights).
// Code size matters here: these kernels run one wavefront per SIMD through long
// straight-line adders (an inlined XYZZ add is ~3.5 k instructions), so every
// extra inline site is instruction-cache misses on the critical path.  Each loop
// below therefore has ONE adder site, operands picked by (uniform) selects.
__device__ __forceinline__ G1Xyzz29 xyzz29_sel(bool c, const G1Xyzz29& a, const G1Xyzz29& b) {
  G1Xyzz29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    r.x.v[i] = c ? a.x.v[i] : b.x.v[i];
    r.y.v[i] = c ? a.y.v[i] : b.y.v[i];
    r.zz.v[i] = c ? a.zz.v[i] : b.zz.v[i];
    r.zzz.v[i] = c ? a.zzz.v[i] : b.zzz.v[i];
  }
  return r;
}

template <bool CAREFUL>
__device__ __forceinline__ bool chunk_sums(const G1Xyzz29* __restrict__ bw, uint32_t base, uint32_t top,
                                           G1Xyzz29& run, G1Xyzz29& acc) {
  run = xyzz29_identity();
  acc = xyzz29_identity();
  bool bad = false;
#pragma unroll 1
  for (uint32_t op = 0; op < 2 * (top - base); ++op) {  // run += B_i ; acc += run   (i descending)
    bool second = op & 1u;
    G1Xyzz29 x = second ? acc : run;
    G1Xyzz29 y = second ? run : bw[top - 1 - (op >> 1)];
    if (CAREFUL) xyzz29_add_careful(x, y);
    else xyzz29_add_skipid_fast(x, y, bad);
    if (second) acc = x;
    else run = x;
  }
  if (CAREFUL) return false;
  // any degenerate intermediate poisons everything downstream of it
  return bad || (!xyzz29_is_identity(run) && xyzz29_is_degenerate(run)) ||
         (!xyzz29_is_identity(acc) && xyzz29_is_degenerate(acc));
}

// 64-lane workgroup fold.  Lane l brings (run_l, acc_l, extra_l); returns (to every lane)
//   out   = sum_l [ extra_l + 2^log2_scale * ( acc_l + 2^log2_l * l * run_l ) ]
//   total = sum_l run_l
// as 14 steps of  x = 2^k x + y :  6 suffix-scan steps (y = the value d lanes up),
// the two weighting steps, 6 more scan steps whose lane 0 ends with the sum.
// Careful adders throughout: partial sums of neighbouring lanes can coincide or
// cancel (duplicated points), and 14 adds per wave are not worth a redo path.
// (`total` goes straight to `total_dst` -- lane 0 stores it when it exists, null: not wanted -- instead of living through
// the remaining eight steps: the compiler kept those 144 bytes on the stack.)
__device__ __forceinline__ void wave_weighted_fold(G1Xyzz29* sh, const G1Xyzz29& run, const G1Xyzz29& acc,
                                                   const G1Xyzz29& extra, int log2_l, int log2_scale,
                                                   G1Xyzz29& out, G1Xyzz29* total_dst) {
  const uint32_t lane = threadIdx.x;
  G1Xyzz29 x = run;
#pragma unroll 1
  for (int step = 0; step < 14; ++step) {
    sh[lane] = x;
    __syncthreads();
    G1Xyzz29 y;
    int ndbl = 0;
    if (step == 6) {
      if (lane == 0 && total_dst) *total_dst = sh[0];
      x = xyzz29_sel(lane >= 1, x, xyzz29_identity());
      y = acc;
      ndbl = log2_l;
    } else if (step == 7) {
      y = extra;
      ndbl = log2_scale;
    } else {
      uint32_t d = 1u << (step < 6 ? step : step - 8);
      y = xyzz29_identity();
      if (lane + d < 64) y = sh[lane + d];
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < ndbl; ++k)
      if (!xyzz29_is_identity(x)) x = xyzz29_double(x);
    xyzz29_add_careful(x, y);
  }
  sh[lane] = x;
  __syncthreads();
  out = sh[0];
}

constexpr int ilog2_const(uint32_t v) { return v <= 1 ? 0 : 1 + ilog2_const(v >> 1); }
static_assert((kChunk & (kChunk - 1)) == 0, "kChunk must be a power of two (chunk weights are applied by doublings)");
constexpr int kLog2Chunk = ilog2_const(kChunk);
constexpr int kLog2BlockBuckets = 6 + kLog2Chunk;  // a P6 block covers 64 * kChunk buckets

// One 64-lane block = 64 consecutive chunks of one window ->
//   block_parts[2 * blk]     = T = sum (i - first + 1) B_i over the block's buckets (first = its first bucket)
//   block_parts[2 * blk + 1] = S = sum B_i
__global__ void __launch_bounds__(64)
    k_bucket_reduce(const G1Xyzz29* __restrict__ buckets, G1Xyzz29* __restrict__ block_parts, PipParams p,
                    uint32_t chunks_per_window, uint32_t blocks_per_window) {
  SNARKV_RAISE_PRIO();
  __shared__ G1Xyzz29 sh[64];
  uint32_t w = blockIdx.x / blocks_per_window, bj = blockIdx.x % blocks_per_window;
  uint32_t j = bj * 64 + threadIdx.x;
  G1Xyzz29 run = xyzz29_identity(), acc = xyzz29_identity();
  if (j < chunks_per_window) {
    const uint32_t chunk = 1u << p.chunk_log2;
    uint32_t base = j * chunk;
    uint32_t top = base + chunk < p.B ? base + chunk : p.B;
    const G1Xyzz29* bw = buckets + (size_t)w * p.B;
    if (chunk_sums<false>(bw, base, top, run, acc)) chunk_sums<true>(bw, base, top, run, acc);
  }
  G1Xyzz29 t;
  wave_weighted_fold(sh, run, acc, xyzz29_identity(), (int)p.chunk_log2, 0, t, &block_parts[2 * (size_t)blockIdx.x + 1]);
  if (threadIdx.x == 0) block_parts[2 * (size_t)blockIdx.x] = t;
}

// --------------------------------------------------------------- P8
// 2^n * P, P uniform over each aligned 4-lane group: the quad-cooperative Jacobian doubling of g1_29.h
__device__ __forceinline__ G1Xyzz29 xyzz29_double_n_quad(const G1Xyzz29& p, int n) {
  if (n <= 0) return p;
  const uint32_t q = threadIdx.x & 3u;
  Fq29 x = fq29_mul(p.x, p.zz);
  Fq29 y = fq29_mul(fq29_norm(p.y), p.zzz);
  Fq29 z = p.zz;
  for (int k = 0; k < n; ++k) jac29_double_quad(x, y, z, q);
  return jac29_to_xyzz(x, y, z);
}

// One wavefront per window.  Lane l owns `per` consecutive P6 blocks:
//   window sum = sum_blk T_blk + 2^kLog2BlockBuckets * sum_blk blk * S_blk
// with the block-index weights applied by the same suffix-scan fold as in P6,
// then shifted[w] = 2^(c (w + w0)) * (window sum)  (the `result.double()` x c
// of msm.rs:285-287) by the quad-cooperative doubling chain.
__global__ void __launch_bounds__(64)
    k_shift_windows(const G1Xyzz29* __restrict__ block_parts, G1Xyzz29* __restrict__ shifted, PipParams p,
                    uint32_t blocks_per_window) {
  SNARKV_RAISE_PRIO();
  __shared__ G1Xyzz29 sh[64];
  uint32_t w = blockIdx.x, lane = threadIdx.x;
  const G1Xyzz29* src = block_parts + 2 * (size_t)w * blocks_per_window;
  uint32_t per = (blocks_per_window + 63) / 64;  // a power of two (B is), or 1
  uint32_t lo = lane * per, hi = lo + per < blocks_per_window ? lo + per : blocks_per_window;
  G1Xyzz29 run = xyzz29_identity(), acc_s = xyzz29_identity(), acc_t = xyzz29_identity();
  uint32_t cnt = hi > lo ? hi - lo : 0;
#pragma unroll 1
  for (uint32_t op = 0; op < 3 * cnt; ++op) {  // per block i (descending): acc_s += run; run += S_i; acc_t += T_i
    uint32_t i = hi - 1 - op / 3, k = op % 3;  // => acc_s = sum (i - lo) S_i
    G1Xyzz29 x = xyzz29_sel(k == 0, acc_s, xyzz29_sel(k == 1, run, acc_t));
    G1Xyzz29 y = run;
    if (k != 0) y = src[2 * (size_t)i + (k == 1 ? 1 : 0)];
    xyzz29_add_careful(x, y);
    if (k == 0) acc_s = x;
    else if (k == 1) run = x;
    else acc_t = x;
  }
  G1Xyzz29 r;
  wave_weighted_fold(sh, run, acc_s, acc_t, 31 - __clz((int)per), 6 + (int)p.chunk_log2, r, nullptr);  // a P6 block covers 64 chunks
  const uint32_t wl = p.wper ? w % p.wper : w;  // window index inside its own MSM
  if (!xyzz29_is_identity(r)) r = xyzz29_double_n_quad(r, p.c * (int)(wl + p.w0));
  if (lane == 0) shifted[w] = r;
}

// --------------------------------------------------------------- P9
// result = sum_w shifted[w]; 64 lanes, LDS tree, then `to_affine` (or the
// projective partial for the multi-GPU fold).  Also used for the fold itself.
__global__ void __launch_bounds__(64)
    k_final(const G1Xyzz29* __restrict__ parts, uint32_t count, uint32_t* __restrict__ out, int partial_out, uint32_t mont) {
  SNARKV_RAISE_PRIO();
  __shared__ G1Xyzz29 sh[64];
  uint32_t lane = threadIdx.x;
  // batched form: workgroup b folds parts[b * count ..) into out[b]  (a single MSM launches one workgroup)
  parts += (size_t)blockIdx.x * count;
  out += (size_t)blockIdx.x * (partial_out ? sizeof(G1Xyzz29) / 4 : 16);
  G1Xyzz29 acc = xyzz29_identity();
#pragma unroll 1
  for (uint32_t i = lane; i < count; i += 64) xyzz29_add_careful(acc, parts[i]);
  sh[lane] = acc;
  __syncthreads();
  uint32_t s0 = 32;  // lanes >= count hold the identity: start the tree at the live width
  while (s0 > 1 && s0 >= count) s0 >>= 1;
#pragma unroll 1
  for (uint32_t s = s0; s >= 1; s >>= 1) {
    if (lane < s) {
      G1Xyzz29 a = sh[lane];
      xyzz29_add_careful(a, sh[lane + s]);
      sh[lane] = a;
    }
    __syncthreads();
  }
  if (lane == 0) {
    if (partial_out) {
      *reinterpret_cast<G1Xyzz29*>(out) = sh[0];
    } else {
      G1Affine29 r = xyzz29_to_affine(sh[0]);
      uint32_t w[16];
      g1a29_to_words(r, w, mont != 0);
      for (int i = 0; i < 16; ++i) out[i] = w[i];
    }
  }
}

// `jobs` folds in one launch: d_partials = [job][count] partials, d_out64s = [job] affine points
int launch_fold_partials_many(snarkv_ctx* ctx, const void* d_partials, size_t count, size_t jobs, void* d_out64s) {
  hipLaunchKernelGGL(k_final, dim3((uint32_t)jobs), dim3(64), 0, ctx->stream, (const G1Xyzz29*)d_partials, (uint32_t)count,
                     (uint32_t*)d_out64s, 0, ctx->mont ? 1u : 0u);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

int launch_fold_partials(snarkv_ctx* ctx, const void* d_partials, size_t count, void* d_out64, bool partial_out) {
  hipLaunchKernelGGL(k_final, dim3(1), dim3(64), 0, ctx->stream, (const G1Xyzz29*)d_partials, (uint32_t)count,
                     (uint32_t*)d_out64, partial_out ? 1 : 0, ctx->mont ? 1u : 0u);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

// Window size of an n-point MSM: the c that minimises  entries + buckets  in field products -- kHalves n W(c) mixed
// additions of ~10 products, W(c) 2^(c-1) buckets at ~30 (two full additions each in the reduce, whose chains are
// latency- rather than throughput-bound) -- among the sizes whose TOP window is populated: with magnitudes of
// kDigitBits - 1 bits the top window holds t = kDigitBits - 1 - (W - 1) c of them, and a narrow one (t < c - 3) piles
// its whole window's entries into a handful of level-1 keys, each sorted by ONE workgroup (2^19 points at c = 15: t = 7,
// partition + sort 1.4 ms instead of 0.12).  For GLV's 127-bit halves that leaves c = 8, 10, 13, 16.  Measured single-MSM
// latency, 2^16 .. 2^19 points, round-1 rule (c = log2 n - 4, windows balanced) against this: 1.29 / 1.19 / 1.49 / 3.11 ms
// -> 1.15 / 1.19 / 1.36 / 1.67 ms (profiles/r02_sweep_window_bits.txt).
static int default_window_bits(size_t n) {
  int best = 2;
  double best_cost = 0;
  for (int c = 2; c <= 22; ++c) {
    const int W = (kDigitBits + c - 1) / c;
    const int top = kDigitBits - 1 - (W - 1) * c;
    double cost = (double)kHalves * (double)n * W * 10.0 + (double)W * (double)(1u << (c - 1)) * 30.0;
    if (top < c - 3) cost *= 1.6;
    if (c == 2 || cost < best_cost) best = c, best_cost = cost;
  }
  return best;
}

// Smallest window size with the same number of windows: keeps the TOP window
// populated (with GLV c = 18 would leave it one useful bit of a 127-bit half-scalar, i.e.
// one bucket holding half of all entries).
static int balance_window_bits(int c) {
  int W = (kDigitBits + c - 1) / c;
  return (kDigitBits + W - 1) / W;
}

// Entries per k_accumulate lane for ONE MSM at a time.  A lane is a serial chain (~5 us per entry when the SIMD is shared
// three ways), so below ~2^20 points 64-entry runs leave the machine to a few hundred wavefronts that each run for
// 0.33 ms whatever n is (measured: k_accumulate 0.33 ms at 2^16 AND 2^17 points).  Shorter runs = more lanes, at the price
// of more head / tail partials for k_combine: the shortest of 16 / 32 / 64 that keeps the launch within ~1.5 rounds of
// the 3 072 wave slots, and 32 rather than 16 when even 16 would not fill two thirds of them.  Measured single-MSM
// latency at 2^16 / 2^17 / 2^18 / 2^19 points: 1.15 / 1.19 / 1.35 / 1.66 -> 1.03 / 1.13 / 1.14 / 1.46 ms
// (profiles/r02_sweep_run_length.txt).  With several MSMs in flight (hint, batch) the long runs stay.
static uint32_t latency_run_length(uint64_t entries) {
  auto waves = [&](uint64_t run) { return (entries + 64 * run - 1) / (64 * run); };
  if (waves(16) <= 4608) return waves(16) >= 2048 ? 16u : 32u;
  if (waves(32) <= 4608) return 32u;
  return (uint32_t)kRun;
}

int launch_msm_pippenger(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int window_bits,
                         void* d_out, bool partial_out, void* d_buckets_out) {
  return launch_msm_pippenger_phases(ctx, ctx->stream, PIP_PHASE_ALL, d_scalars, d_points, n, window_bits, d_out,
                                     partial_out, d_buckets_out, nullptr);
}

// The same launch cut into its three phases, each enqueued on a stream of the caller's choice (the batch scheduler of
// capi.hip runs the phases of MANY MSMs in phase order):
//   PIP_PHASE_SORT  P0-P4  prepare, scan, partition + sort               (reads the inputs, fills ctx's scratch)
//   PIP_PHASE_ACC   P5     bucket accumulation + combine                 (scratch -> bucket grid)
//   PIP_PHASE_TAIL  P6-P9  bucket reduce, shift chains, final            (bucket grid -> d_out)
// `ctx` owns the scratch (its own stream is not used unless passed as `st`); `d_grid`, when given, is the bucket grid
// to fill in place of the context's own (a batch lays its MSMs' grids end to end for one batched tail).  Every call
// of one MSM must pass the same n / window_bits; stage marks exist only for PIP_PHASE_ALL.
// (Measured and removed: window groups -- the sorted stream accumulated group by group, top windows first, each group's
// tail on a side stream: 1 / 2 / 4 / 8 groups -> single-MSM latency 2.23 / 2.55 / 2.90 / 4.51 ms,
// profiles/r02_sweep_gsz.txt, git tag exp/window-groups.)
int launch_msm_pippenger_phases(snarkv_ctx* ctx, hipStream_t st, int phases, const void* d_scalars, const void* d_points,
                                size_t n, int window_bits, void* d_out, bool partial_out, void* d_buckets_out,
                                void* d_grid) {
  PipParams p;
  p.w0 = 0;
  p.wper = 0;
  p.chunk_log2 = (uint32_t)kLog2Chunk;
  p.mont = ctx->mont ? 1u : 0u;
  p.n = (uint32_t)n;
  p.c = window_bits > 0 ? window_bits : balance_window_bits(default_window_bits(n));
  if (p.c < 2) p.c = 2;
  if (p.c > 22) p.c = 22;
  p.W = (kDigitBits + p.c - 1) / p.c;
  p.B = 1u << (p.c - 1);
  p.nb = (uint32_t)p.W * p.B;
  // level-1 keys: ~kSortTarget items each so that a level-2 workgroup sorts its
  // slice inside LDS; bounded by the LDS counters of a tile workgroup
  int high = p.c - 1 > 10 ? p.c - 1 - 10 : 0;  // at most 1024 level-2 bins (LDS)
  while (high < p.c - 1 && ((kHalves * n) >> (high + 1)) >= kSortTarget && ((uint64_t)p.W << (high + 1)) <= kMaxKeys) ++high;
  p.low_bits = (p.c - 1) - high;
  p.SB = p.B >> p.low_bits;
  p.nkeys = (uint32_t)p.W * p.SB;
  // the staged partition (k_sort_scatter_staged) owns one tile of exactly SNARKV_TILE_BASE scalars per workgroup
  p.tile = SNARKV_TILE_BASE;
  static_assert(kStageItems * kHalves <= 32, "a staged lane keeps one sign bit per entry in a 32-bit word");
  const size_t lds_staged = ((size_t)3 * p.SB + 16 + (p.SB & 1u)) * 4 + (size_t)kStageScalars * kHalves * 8;
  p.nblk = (uint32_t)((n + p.tile - 1) / p.tile);
  p.mstride = p.nblk | 1u;
  uint64_t max_entries = (uint64_t)kHalves * (uint64_t)n * (uint64_t)p.W;
  if (n == 0 || max_entries >= 0xFFFFFFFFull || n >= 0x40000000ull) {
    set_last_error("pippenger: n=%zu out of range", n);
    return SNARKV_ERR_LENGTH;
  }
  if ((size_t)p.nkeys * 4 > 65536 || lds_staged > 96 * 1024) {
    set_last_error("pippenger: key table too large (nkeys=%u)", p.nkeys);
    return SNARKV_ERR_LENGTH;
  }
  p.krun = ctx->throughput_mode ? (uint32_t)kRunThroughput : latency_run_length(max_entries);
  const uint32_t max_runs = (uint32_t)((max_entries + p.krun - 1) / p.krun) + 1;  // run slots (head / tail partial each)
  uint32_t mcount = p.nkeys * p.mstride;
  uint32_t scan_blocks = (mcount + 1023) / 1024;
  uint32_t chunks_per_window = (p.B + kChunk - 1) / kChunk;
  uint32_t blocks_per_window = (chunks_per_window + 63) / 64;

  void *d_pts, *d_glv, *d_counts, *d_offsets, *d_M, *d_blocksum, *d_entries, *d_tmp, *d_seg_ids, *d_seg_parts,
      *d_buckets, *d_wave, *d_shift, *d_misc, *d_big;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_POINTS_MONT, kHalves * n * sizeof(G1Packed), &d_pts));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_GLV, n * 32, &d_glv));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_COUNTS, (size_t)p.nb * 4, &d_counts));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OFFSETS, (size_t)p.nb * 4, &d_offsets));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_CURSOR, (size_t)mcount * 4, &d_M));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_BLOCKSUMS, (size_t)scan_blocks * 4 + 64, &d_blocksum));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_ENTRIES, max_entries * 8, &d_entries));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SORT_TMP, max_entries * 8, &d_tmp));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SEG_IDS, (size_t)max_runs * 8, &d_seg_ids));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SEG_PARTIALS, (size_t)max_runs * 2 * sizeof(G1Xyzz29), &d_seg_parts));
  if (d_grid) d_buckets = d_grid;
  else SNARKV_TRY(ctx_reserve(ctx, SLOT_BUCKETS, (size_t)p.nb * sizeof(G1Xyzz29), &d_buckets));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_CHUNK_PARTIALS, 2 * (size_t)blocks_per_window * p.W * sizeof(G1Xyzz29), &d_wave));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SHIFTED, (size_t)p.W * sizeof(G1Xyzz29), &d_shift));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_MISC, 64, &d_misc));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_BIG_LIST, (size_t)kMaxBig * 4, &d_big));
  uint32_t* d_total = (uint32_t*)d_misc;  // [0] entries of the sorted stream, [4] big-bucket counter of k_combine
  uint32_t* d_big_count = d_total + 4;

  const bool tm = ctx->stage_timing && phases == PIP_PHASE_ALL;
  const bool tm_acc = ctx->stage_timing && phases == PIP_PHASE_ACC;  // a batch times its accumulations: ev[3] .. ev[4] .. ev[5]
  int evi = 0;
#define STAGE_MARK()                                         \
  do {                                                       \
    if (tm) SNARKV_HIP(hipEventRecord(ctx->ev[evi++], st));  \
  } while (0)
  if ((tm || tm_acc) && !ctx->ev_ready) {
    for (int i = 0; i <= SNARKV_PIP_STAGES; ++i) SNARKV_HIP(hipEventCreate(&ctx->ev[i]));
    ctx->ev_ready = true;
  }
  STAGE_MARK();  // 0
  if (phases & PIP_PHASE_SORT) {
    // (no hipMemsetAsync anywhere: the matrix' padding column, the counters and the empty buckets are filled by
    // k_prepare / k_sort_level2 / k_combine themselves -- 60 fill launches per 20-job batch less, level in time)
    hipLaunchKernelGGL(k_prepare, dim3(p.nblk), dim3(SNARKV_PREP_THREADS), (size_t)p.nkeys * 4, st, (const uint32_t*)d_scalars,
                       (const uint32_t*)d_points, (G1Packed*)d_pts, (uint4*)d_glv, p, (uint32_t*)d_M);
    STAGE_MARK();  // 1: prepare (GLV split, phi(P), to Montgomery) + digit histogram
    hipLaunchKernelGGL(k_scan_local, dim3(scan_blocks), dim3(256), 0, st, (uint32_t*)d_M, (uint32_t*)d_blocksum, mcount);
    hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(1024), 0, st, (uint32_t*)d_blocksum, scan_blocks, d_total);
    hipLaunchKernelGGL(k_scan_add, dim3(scan_blocks), dim3(256), 0, st, (uint32_t*)d_M, (const uint32_t*)d_blocksum,
                       mcount);
    STAGE_MARK();  // 2: scan
    static std::atomic<uint64_t> attr_set{0};  // per device: more dynamic LDS than the 64 KiB default
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
      SNARKV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_scatter_staged),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(k_sort_scatter_staged, dim3(p.nblk), dim3(SNARKV_TILE_THREADS), lds_staged, st, (const uint4*)d_glv,
                       p, (const uint32_t*)d_M, (uint2*)d_tmp);
    const uint32_t nbins_l2 = 1u << p.low_bits;
    const size_t lds2 = ((size_t)nbins_l2 + SNARKV_L2_THREADS + 1) * 4 + (size_t)kSortCap * 8;  // < 64 KiB: nbins <= 1 024
    hipLaunchKernelGGL(k_sort_level2, dim3(p.nkeys), dim3(SNARKV_L2_THREADS), lds2, st, (const uint2*)d_tmp, (const uint32_t*)d_M,
                       d_total, p, (uint2*)d_entries, (uint32_t*)d_counts, (uint32_t*)d_offsets);
  }
  STAGE_MARK();  // 3: partition + level-2 sort
  // (Scheduling experiments measured and dropped -- a shared accumulate stream across in-flight MSMs, s_setprio on / off,
  // an occupancy cap on k_accumulate: DESIGN.md section 4.  The plateau is total issued work, not scheduling.)
  if (tm_acc) SNARKV_HIP(hipEventRecord(ctx->ev[3], st));
  if (phases & PIP_PHASE_ACC) {
    auto acc_kernel = p.krun == 16u ? k_accumulate<16> : p.krun == 32u ? k_accumulate<32>
                      : p.krun == (uint32_t)kRun ? k_accumulate<kRun> : k_accumulate<kRunThroughput>;
    hipLaunchKernelGGL(acc_kernel, dim3((max_runs + 63) / 64), dim3(64), 0, st, (const uint2*)d_entries,
                       (const uint32_t*)d_total, (const G1Packed*)d_pts, (G1Xyzz29*)d_buckets, (uint32_t*)d_seg_ids,
                       (G1Xyzz29*)d_seg_parts);
  }
  STAGE_MARK();  // 4: bucket accumulate
  if (tm_acc) SNARKV_HIP(hipEventRecord(ctx->ev[4], st));
  if (phases & PIP_PHASE_ACC) {
    hipLaunchKernelGGL(k_combine, dim3((p.nb + 63) / 64), dim3(64), 0, st, (const uint32_t*)d_counts,
                       (const uint32_t*)d_offsets, p, (const uint2*)d_entries, (const G1Packed*)d_pts,
                       (const uint32_t*)d_seg_ids, (const G1Xyzz29*)d_seg_parts, (G1Xyzz29*)d_buckets, d_big_count,
                       (uint32_t*)d_big);
    // one workgroup per oversized bucket; idle workgroups exit at once
    const uint32_t big_grid = std::min<uint32_t>(max_runs / kBigSpan + 1, kBigGrid);
    hipLaunchKernelGGL(k_combine_big, dim3(big_grid), dim3(256), 0, st, (const uint32_t*)d_counts,
                       (const uint32_t*)d_offsets, (const uint2*)d_entries, (const G1Packed*)d_pts,
                       (const uint32_t*)d_seg_ids, (const G1Xyzz29*)d_seg_parts, (G1Xyzz29*)d_buckets,
                       (const uint32_t*)d_big_count, (const uint32_t*)d_big, p);
    if (tm_acc) SNARKV_HIP(hipEventRecord(ctx->ev[5], st));
  }
  STAGE_MARK();  // 5: bucket combine
  if (d_buckets_out) {  // bucket-sharded variant: hand the (sanitised) bucket sums out and stop here
    if (phases & PIP_PHASE_ACC)
      SNARKV_HIP(hipMemcpyAsync(d_buckets_out, d_buckets, (size_t)p.nb * sizeof(G1Xyzz29), hipMemcpyDeviceToDevice, st));
    SNARKV_HIP(hipGetLastError());
    return SNARKV_OK;
  }
  if (phases & PIP_PHASE_TAIL)
    hipLaunchKernelGGL(k_bucket_reduce, dim3(blocks_per_window * p.W), dim3(64), 0, st, (const G1Xyzz29*)d_buckets,
                       (G1Xyzz29*)d_wave, p, chunks_per_window, blocks_per_window);
  STAGE_MARK();  // 6: bucket reduce
  if (phases & PIP_PHASE_TAIL)
    hipLaunchKernelGGL(k_shift_windows, dim3(p.W), dim3(64), 0, st, (const G1Xyzz29*)d_wave, (G1Xyzz29*)d_shift, p,
                       blocks_per_window);
  STAGE_MARK();  // 7: window sums + 2^(cw) shift chains
  if (phases & PIP_PHASE_TAIL)
    hipLaunchKernelGGL(k_final, dim3(1), dim3(64), 0, st, (const G1Xyzz29*)d_shift, (uint32_t)p.W, (uint32_t*)d_out,
                       partial_out ? 1 : 0, p.mont);
  STAGE_MARK();  // 8: final sum + to_affine
#undef STAGE_MARK
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

// ---- bucket-sharded variant ("bucket-sum allreduce") ---------------------------
// Every GPU fills the GLOBAL bucket grid (same c for all) from its point shard; the
// grids are exchanged by window range and summed; each GPU reduces the windows it owns.
int pip_geometry(size_t n_total, int window_bits, uint32_t* c, uint32_t* windows, uint32_t* buckets_per_window) {
  int cc = window_bits > 0 ? window_bits : balance_window_bits(default_window_bits(n_total));
  if (cc < 2) cc = 2;
  if (cc > 22) cc = 22;
  *c = (uint32_t)cc;
  *windows = (uint32_t)((kDigitBits + cc - 1) / cc);
  *buckets_per_window = 1u << (cc - 1);
  return SNARKV_OK;
}

__global__ void __launch_bounds__(256) k_buckets_add(G1Xyzz29* __restrict__ dst, const G1Xyzz29* __restrict__ src, uint32_t count) {
  SNARKV_RAISE_PRIO();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  G1Xyzz29 a = dst[i];
  xyzz29_add_careful(a, src[i]);  // the same point may sit in both shards' buckets: doubling / cancellation handled
  dst[i] = xyzz29_sanitize(a);
}

int launch_buckets_add(snarkv_ctx* ctx, void* d_dst, const void* d_src, size_t count) {
  hipLaunchKernelGGL(k_buckets_add, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, ctx->stream, (G1Xyzz29*)d_dst,
                     (const G1Xyzz29*)d_src, (uint32_t)count);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

// sum_{w in [w0, w0 + wcount)} 2^(c w) * sum_b b * buckets[w - w0][b]  ->  one projective partial
int launch_buckets_reduce(snarkv_ctx* ctx, const void* d_buckets, uint32_t c, uint32_t w0, uint32_t wcount,
                          void* d_partial) {
  PipParams p;
  memset(&p, 0, sizeof p);
  p.c = (int)c;
  p.W = (int)wcount;
  p.B = 1u << (c - 1);
  p.nb = wcount * p.B;
  p.w0 = w0;
  p.chunk_log2 = (uint32_t)kLog2Chunk;
  uint32_t chunks_per_window = (p.B + kChunk - 1) / kChunk;
  uint32_t blocks_per_window = (chunks_per_window + 63) / 64;
  void *d_wave, *d_shift;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_CHUNK_PARTIALS, 2 * (size_t)blocks_per_window * wcount * sizeof(G1Xyzz29), &d_wave));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SHIFTED, (size_t)wcount * sizeof(G1Xyzz29), &d_shift));
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_bucket_reduce, dim3(blocks_per_window * wcount), dim3(64), 0, st, (const G1Xyzz29*)d_buckets,
                     (G1Xyzz29*)d_wave, p, chunks_per_window, blocks_per_window);
  hipLaunchKernelGGL(k_shift_windows, dim3(wcount), dim3(64), 0, st, (const G1Xyzz29*)d_wave, (G1Xyzz29*)d_shift, p,
                     blocks_per_window);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(64), 0, st, (const G1Xyzz29*)d_shift, wcount, (uint32_t*)d_partial, 1, 0u);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

// The tail of `jobs` MSMs of the same geometry whose bucket grids lie end to end ([job][window][bucket]): three
// launches for all of them (a tail is latency-bound -- 14-step folds and the 2^(c w) doubling chains -- so `jobs` of
// them cost what one does), out[job] = the affine sum (64 B) or the projective partial.
// geometry + scratch of a batched tail over `jobs` grids
struct TailGeom {
  PipParams p;
  uint32_t chunks_per_window, blocks_per_window;
  void *d_wave, *d_shift;
};
static int tail_geometry(snarkv_ctx* ctx, uint32_t c, uint32_t windows, uint32_t jobs, TailGeom* g) {
  PipParams& p = g->p;
  memset(&p, 0, sizeof p);
  p.c = (int)c;
  p.W = (int)(windows * jobs);
  p.B = 1u << (c - 1);
  p.nb = windows * jobs * p.B;
  p.w0 = 0;
  p.wper = windows;
  const uint32_t wtotal = windows * jobs;
  // Buckets per k_bucket_reduce lane.  One MSM's reduce is a latency chain on a few hundred wavefronts: short chunks (8
  // buckets: 16 serial additions + the 14-step fold) keep it short.  A batch's tail reduces `jobs` grids -- thousands of
  // wavefronts, throughput-bound -- where the fold is 14 of every 30 additions: chunks of 32 do 78 additions per 32
  // buckets instead of 120.  Same bytes for any chunk size.
  // The chunk is picked by that model: rounds of the machine's wave slots (194 VGPRs: two wavefronts per SIMD) x the
  // additions of one wavefront, 2 chunk + 14 -- 16 buckets per lane for 20 or 40 grids of 8 x 32 768 buckets, 8 for one.
  uint32_t cl2 = (uint32_t)kLog2Chunk;
  {
    static int slots = 0;  // 2 wavefronts x 4 SIMDs x CUs
    if (!slots) {
      hipDeviceProp_t prop;
      slots = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0 ? 8 * prop.multiProcessorCount : 2048;
    }
    uint64_t best = ~0ull;
    for (uint32_t q = (uint32_t)kLog2Chunk; q <= 6; ++q) {
      const uint64_t blocks = ((uint64_t)(p.B >> q) + 63) / 64, waves = (uint64_t)jobs * windows * (blocks ? blocks : 1);
      const uint64_t cost = ((waves + slots - 1) / slots) * (2ull * (1u << q) + 14);
      if (cost < best) best = cost, cl2 = q;
    }
  }
  while ((1u << cl2) > p.B && cl2 > 0) --cl2;
  p.chunk_log2 = cl2;
  const uint32_t chunk = 1u << cl2;
  g->chunks_per_window = (p.B + chunk - 1) / chunk;
  g->blocks_per_window = (g->chunks_per_window + 63) / 64;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_CHUNK_PARTIALS, 2 * (size_t)g->blocks_per_window * wtotal * sizeof(G1Xyzz29), &g->d_wave));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SHIFTED, (size_t)wtotal * sizeof(G1Xyzz29), &g->d_shift));
  return SNARKV_OK;
}

// The tail of `jobs` MSMs of the same geometry whose bucket grids lie end to end ([job][window][bucket]): three
// launches for all of them, out[job] = the affine sum (64 B) or the projective partial.
// (Measured and removed: every job's bucket reduce behind its own combine, under the other jobs' accumulations --
// 1-14 % slower, profiles/r03_ab_scheduling.txt, git tag exp/many-tail.)
int launch_buckets_reduce_many(snarkv_ctx* ctx, hipStream_t st, const void* d_grids, uint32_t c, uint32_t windows,
                               uint32_t jobs, void* d_out, bool partial_out) {
  TailGeom g;
  SNARKV_TRY(tail_geometry(ctx, c, windows, jobs, &g));
  const uint32_t wtotal = windows * jobs;
  hipLaunchKernelGGL(k_bucket_reduce, dim3(g.blocks_per_window * wtotal), dim3(64), 0, st, (const G1Xyzz29*)d_grids,
                     (G1Xyzz29*)g.d_wave, g.p, g.chunks_per_window, g.blocks_per_window);
  hipLaunchKernelGGL(k_shift_windows, dim3(wtotal), dim3(64), 0, st, (const G1Xyzz29*)g.d_wave, (G1Xyzz29*)g.d_shift, g.p,
                     g.blocks_per_window);
  hipLaunchKernelGGL(k_final, dim3(jobs), dim3(64), 0, st, (const G1Xyzz29*)g.d_shift, windows, (uint32_t*)d_out,
                     partial_out ? 1 : 0, ctx->mont ? 1u : 0u);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
