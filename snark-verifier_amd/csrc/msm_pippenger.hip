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
//   P0 k_to_mont        points: canonical LE -> Montgomery, once (HBM-resident copy)
//   P1 k_digit_count    signed c-bit digits (half the buckets of msm.rs:269),
//                       histogram of (window, |digit|) with device atomics
//   P2 k_scan_*         exclusive scan of the histogram -> bucket offsets
//   P3 k_digit_scatter  counting-sort scatter of (bucket, point index, sign)
//   P4 k_accumulate     every lane owns a FIXED-LENGTH run of the sorted
//                       stream (S entries) -- perfectly load-balanced whatever
//                       the scalar distribution -- and emits a head partial, a
//                       tail partial and complete interior buckets
//                       (the `buckets[d-1].add_assign(base)` of msm.rs:291-296)
//   P5 k_combine        per bucket: stitch the partials of the runs it spans
//   P6 k_bucket_reduce  running-sum trick of msm.rs:298-302, chunked so that
//                       >= 64 K lanes work; each chunk's sum is weighted by its
//                       base index with a short double-and-add
//   P7 k_sum_groups     per-window tree sum of the chunk partials (LDS)
//   P8 k_window_fold    Horner over windows (the `result.double()` x c of
//                       msm.rs:285-287) + `to_affine`, or the projective
//                       partial for the multi-GPU fold.
//
// MFMA is deliberately unused: there is no dense contraction, the work is
// 254-bit modular multiplication on the integer VALU (v_mad_u64_u32).
#include "ctx.hpp"
#include "g1.cuh"

namespace snarkv {

constexpr int kRun = 32;         // P4: entries per lane
constexpr int kChunk = 8;        // P6: buckets per lane
constexpr uint32_t kNoBucket = 0xFFFFFFFFu;

struct PipParams {
  uint32_t n;
  int c;          // window bits
  int W;          // windows = ceil(255 / c)
  uint32_t B;     // buckets per window = 2^(c-1)  (signed digits)
  uint32_t nb;    // W * B
};

__device__ __forceinline__ uint32_t scalar_bits(const uint32_t* __restrict__ k, int lo, int c) {
  if (lo >= 256) return 0;
  int word = lo >> 5, sh = lo & 31;
  uint64_t v = k[word];
  if (word + 1 < 8) v |= (uint64_t)k[word + 1] << 32;
  return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// --------------------------------------------------------------- P0
__global__ void k_to_mont(const uint32_t* __restrict__ points, G1Affine* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
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
  out[i] = g1a_from_canonical(w);
}

// --------------------------------------------------------------- P1 / P3
// Signed-digit recoding: raw = bits + carry in [0, 2^c]; raw > 2^(c-1) becomes
// raw - 2^c (negative) with a carry into the next window.  With W*c >= 255 and
// scalars < r < 2^254 the top window never carries out.
template <bool SCATTER>
__global__ void k_digits(const uint32_t* __restrict__ scalars, PipParams p, uint32_t* __restrict__ counts_or_cursor,
                         uint2* __restrict__ entries) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const uint32_t* k = scalars + (size_t)i * 8;
  uint32_t carry = 0;
  for (int w = 0; w < p.W; ++w) {
    uint32_t raw = scalar_bits(k, w * p.c, p.c) + carry;
    uint32_t neg = raw > p.B ? 1u : 0u;
    uint32_t d = neg ? ((1u << p.c) - raw) : raw;
    carry = neg;
    if (d != 0) {
      uint32_t b = (uint32_t)w * p.B + d - 1;
      if (SCATTER) {
        uint32_t pos = atomicAdd(&counts_or_cursor[b], 1u);
        entries[pos] = make_uint2(b, i | (neg << 31));
      } else {
        atomicAdd(&counts_or_cursor[b], 1u);
      }
    }
  }
}

// --------------------------------------------------------------- P2
// Exclusive scan of `nb` counters: 1024 per block (256 lanes x 4), block sums
// scanned by one block, then added back.  offsets[nb] = total.
__global__ void __launch_bounds__(256) k_scan_local(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                    uint32_t* __restrict__ blocksum, uint32_t nb) {
  __shared__ uint32_t sh[256];
  uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
  uint32_t v[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = (base + j < nb) ? in[base + j] : 0;
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
    if (base + j < nb) out[base + j] = excl;
    excl += v[j];
  }
  if (threadIdx.x == 255) blocksum[blockIdx.x] = sh[255];
}

__global__ void __launch_bounds__(1024) k_scan_blocksums(uint32_t* __restrict__ blocksum, uint32_t nblocks,
                                                         uint32_t* __restrict__ total_out) {
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

__global__ void __launch_bounds__(256) k_scan_add(uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                                                  const uint32_t* __restrict__ blocksum, uint32_t nb) {
  uint32_t base = blockIdx.x * 1024 + threadIdx.x * 4;
  uint32_t add = blocksum[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < nb) {
      uint32_t v = offsets[base + j] + add;
      offsets[base + j] = v;
      cursor[base + j] = v;
    }
}

// --------------------------------------------------------------- P4
__global__ void __launch_bounds__(64)
    k_accumulate(const uint2* __restrict__ entries, const uint32_t* __restrict__ total_ptr,
                 const G1Affine* __restrict__ pts, G1Xyzz* __restrict__ buckets, uint32_t* __restrict__ seg_ids,
                 G1Xyzz* __restrict__ seg_parts) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t total = *total_ptr;
  uint64_t begin64 = (uint64_t)t * kRun;
  if (begin64 >= total) return;
  uint32_t begin = (uint32_t)begin64;
  uint32_t end = (total - begin > (uint32_t)kRun) ? begin + kRun : total;
  uint32_t cur = entries[begin].x;
  bool first = true;
  G1Xyzz acc = xyzz_identity();
  for (uint32_t e = begin; e < end; ++e) {
    uint2 ent = entries[e];
    if (ent.x != cur) {
      if (first) {
        seg_ids[2 * (size_t)t] = cur;
        seg_parts[2 * (size_t)t] = acc;
        first = false;
      } else {
        buckets[cur] = acc;  // complete interior bucket
      }
      cur = ent.x;
      acc = xyzz_identity();
    }
    G1Affine p = pts[ent.y & 0x7FFFFFFFu];
    if (ent.y >> 31) p.y = fq_neg(p.y);
    xyzz_add_mixed(acc, p);
  }
  if (first) {
    seg_ids[2 * (size_t)t] = cur;
    seg_parts[2 * (size_t)t] = acc;
    seg_ids[2 * (size_t)t + 1] = kNoBucket;
  } else {
    seg_ids[2 * (size_t)t + 1] = cur;
    seg_parts[2 * (size_t)t + 1] = acc;
  }
}

// --------------------------------------------------------------- P5
__global__ void __launch_bounds__(64)
    k_combine(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ offsets,
              const uint32_t* __restrict__ seg_ids, const G1Xyzz* __restrict__ seg_parts,
              G1Xyzz* __restrict__ buckets, uint32_t nb) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  uint32_t cnt = counts[b];
  if (cnt == 0) return;  // bucket array was zero-filled = identity
  uint32_t o = offsets[b];
  uint32_t t0 = o / kRun, t1 = (o + cnt - 1) / kRun;
  G1Xyzz acc = xyzz_identity();
  bool touched = false;
  for (uint32_t t = t0; t <= t1; ++t) {
    if (seg_ids[2 * (size_t)t] == b) {
      xyzz_add(acc, seg_parts[2 * (size_t)t]);
      touched = true;
    }
    if (seg_ids[2 * (size_t)t + 1] == b) {
      xyzz_add(acc, seg_parts[2 * (size_t)t + 1]);
      touched = true;
    }
  }
  if (touched) buckets[b] = acc;
}

// --------------------------------------------------------------- P6
// Lane (w, j) folds buckets [j*kChunk, (j+1)*kChunk) of window w:
//   run = sum B_i ;  acc = sum (i - base + 1) B_i   (running-sum trick)
//   partial = acc + base * run,   base = j*kChunk   (bucket i has weight i+1)
__global__ void __launch_bounds__(64)
    k_bucket_reduce(const G1Xyzz* __restrict__ buckets, G1Xyzz* __restrict__ chunk_parts, PipParams p,
                    uint32_t chunks_per_window) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= chunks_per_window * (uint32_t)p.W) return;
  uint32_t w = g / chunks_per_window, j = g % chunks_per_window;
  uint32_t base = j * kChunk;
  uint32_t top = base + kChunk < p.B ? base + kChunk : p.B;
  const G1Xyzz* bw = buckets + (size_t)w * p.B;
  G1Xyzz run = xyzz_identity(), acc = xyzz_identity();
  for (uint32_t i = top; i-- > base;) {
    xyzz_add(run, bw[i]);
    xyzz_add(acc, run);
  }
  // base * run by double-and-add over the (<= 20) bits of base
  G1Xyzz m = xyzz_identity();
  for (int bit = 31 - __clz((int)(base | 1u)); bit >= 0; --bit) {
    m = xyzz_double(m);
    if ((base >> bit) & 1u) xyzz_add(m, run);
  }
  xyzz_add(acc, m);
  chunk_parts[g] = acc;
}

// --------------------------------------------------------------- P7
// out[g] = sum of in[g*group .. (g+1)*group): 256 lanes stride + LDS tree.
__global__ void __launch_bounds__(256)
    k_sum_groups(const G1Xyzz* __restrict__ in, G1Xyzz* __restrict__ out, uint32_t group) {
  __shared__ G1Xyzz sh[256];
  const G1Xyzz* src = in + (size_t)blockIdx.x * group;
  G1Xyzz acc = xyzz_identity();
  for (uint32_t i = threadIdx.x; i < group; i += 256) xyzz_add(acc, src[i]);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = 128; s >= 1; s >>= 1) {
    if (threadIdx.x < s) {
      G1Xyzz a = sh[threadIdx.x];
      xyzz_add(a, sh[threadIdx.x + s]);
      sh[threadIdx.x] = a;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// --------------------------------------------------------------- P8
__global__ void __launch_bounds__(64)
    k_window_fold(const G1Xyzz* __restrict__ window_sums, PipParams p, uint32_t* __restrict__ out, int partial_out) {
  if (threadIdx.x != 0) return;
  G1Xyzz r = xyzz_identity();
  for (int w = p.W - 1; w >= 0; --w) {
    for (int k = 0; k < p.c; ++k) r = xyzz_double(r);
    xyzz_add(r, window_sums[w]);
  }
  if (partial_out) {
    *reinterpret_cast<G1Xyzz*>(out) = r;
  } else {
    G1Affine a = xyzz_to_affine(r);
    uint32_t w16[16];
    g1a_to_canonical(a, w16);
    for (int i = 0; i < 16; ++i) out[i] = w16[i];
  }
}

static int default_window_bits(size_t n) {
  int lg = 0;
  while (((size_t)1 << (lg + 1)) <= n) ++lg;
  int c = lg - 4;
  if (c < 2) c = 2;
  if (c > 20) c = 20;
  return c;
}

int launch_msm_pippenger(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int window_bits,
                         void* d_out, bool partial_out) {
  PipParams p;
  p.n = (uint32_t)n;
  p.c = window_bits > 0 ? window_bits : default_window_bits(n);
  if (p.c < 2) p.c = 2;
  if (p.c > 22) p.c = 22;
  p.W = (255 + p.c - 1) / p.c;
  p.B = 1u << (p.c - 1);
  p.nb = (uint32_t)p.W * p.B;
  uint64_t max_entries = (uint64_t)n * (uint64_t)p.W;
  if (n == 0 || max_entries >= 0xFFFFFFFFull || n >= 0x80000000ull) {
    set_last_error("pippenger: n=%zu out of range", n);
    return SNARKV_ERR_LENGTH;
  }
  uint32_t nruns = (uint32_t)((max_entries + kRun - 1) / kRun);
  uint32_t scan_blocks = (p.nb + 1023) / 1024;
  uint32_t chunks_per_window = (p.B + kChunk - 1) / kChunk;

  void *d_pts, *d_counts, *d_offsets, *d_cursor, *d_blocksum, *d_entries, *d_seg_ids, *d_seg_parts, *d_buckets,
      *d_chunk, *d_wsum, *d_misc;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_POINTS_MONT, n * sizeof(G1Affine), &d_pts));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_COUNTS, (size_t)p.nb * 4, &d_counts));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OFFSETS, (size_t)p.nb * 4, &d_offsets));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_CURSOR, (size_t)p.nb * 4, &d_cursor));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_BLOCKSUMS, (size_t)scan_blocks * 4 + 64, &d_blocksum));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_ENTRIES, max_entries * 8, &d_entries));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SEG_IDS, (size_t)nruns * 8, &d_seg_ids));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SEG_PARTIALS, (size_t)nruns * 2 * sizeof(G1Xyzz), &d_seg_parts));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_BUCKETS, (size_t)p.nb * sizeof(G1Xyzz), &d_buckets));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_CHUNK_PARTIALS, (size_t)chunks_per_window * p.W * sizeof(G1Xyzz), &d_chunk));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_WINDOW_SUMS, (size_t)p.W * sizeof(G1Xyzz), &d_wsum));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_MISC, 64, &d_misc));
  uint32_t* d_total = (uint32_t*)d_misc;

  hipStream_t st = ctx->stream;
  bool tm = ctx->stage_timing;
  int evi = 0;
#define STAGE_MARK()                                         \
  do {                                                       \
    if (tm) SNARKV_HIP(hipEventRecord(ctx->ev[evi++], st));  \
  } while (0)
  if (tm && !ctx->ev_ready) {
    for (int i = 0; i <= SNARKV_PIP_STAGES; ++i) SNARKV_HIP(hipEventCreate(&ctx->ev[i]));
    ctx->ev_ready = true;
  }
  STAGE_MARK();  // 0
  hipLaunchKernelGGL(k_to_mont, dim3((p.n + 255) / 256), dim3(256), 0, st, (const uint32_t*)d_points,
                     (G1Affine*)d_pts, p.n);
  STAGE_MARK();  // 1
  SNARKV_HIP(hipMemsetAsync(d_counts, 0, (size_t)p.nb * 4, st));
  hipLaunchKernelGGL(k_digits<false>, dim3((p.n + 255) / 256), dim3(256), 0, st, (const uint32_t*)d_scalars, p,
                     (uint32_t*)d_counts, (uint2*)nullptr);
  STAGE_MARK();  // 2
  hipLaunchKernelGGL(k_scan_local, dim3(scan_blocks), dim3(256), 0, st, (const uint32_t*)d_counts,
                     (uint32_t*)d_offsets, (uint32_t*)d_blocksum, p.nb);
  hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(1024), 0, st, (uint32_t*)d_blocksum, scan_blocks, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3(scan_blocks), dim3(256), 0, st, (uint32_t*)d_offsets, (uint32_t*)d_cursor,
                     (const uint32_t*)d_blocksum, p.nb);
  STAGE_MARK();  // 3
  hipLaunchKernelGGL(k_digits<true>, dim3((p.n + 255) / 256), dim3(256), 0, st, (const uint32_t*)d_scalars, p,
                     (uint32_t*)d_cursor, (uint2*)d_entries);
  STAGE_MARK();  // 4
  SNARKV_HIP(hipMemsetAsync(d_buckets, 0, (size_t)p.nb * sizeof(G1Xyzz), st));
  hipLaunchKernelGGL(k_accumulate, dim3((nruns + 63) / 64), dim3(64), 0, st, (const uint2*)d_entries,
                     (const uint32_t*)d_total, (const G1Affine*)d_pts, (G1Xyzz*)d_buckets, (uint32_t*)d_seg_ids,
                     (G1Xyzz*)d_seg_parts);
  STAGE_MARK();  // 5
  hipLaunchKernelGGL(k_combine, dim3((p.nb + 63) / 64), dim3(64), 0, st, (const uint32_t*)d_counts,
                     (const uint32_t*)d_offsets, (const uint32_t*)d_seg_ids, (const G1Xyzz*)d_seg_parts,
                     (G1Xyzz*)d_buckets, p.nb);
  STAGE_MARK();  // 6
  hipLaunchKernelGGL(k_bucket_reduce, dim3((chunks_per_window * p.W + 63) / 64), dim3(64), 0, st,
                     (const G1Xyzz*)d_buckets, (G1Xyzz*)d_chunk, p, chunks_per_window);
  STAGE_MARK();  // 7
  hipLaunchKernelGGL(k_sum_groups, dim3(p.W), dim3(256), 0, st, (const G1Xyzz*)d_chunk, (G1Xyzz*)d_wsum,
                     chunks_per_window);
  hipLaunchKernelGGL(k_window_fold, dim3(1), dim3(64), 0, st, (const G1Xyzz*)d_wsum, p, (uint32_t*)d_out,
                     partial_out ? 1 : 0);
  STAGE_MARK();  // 8
#undef STAGE_MARK
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
