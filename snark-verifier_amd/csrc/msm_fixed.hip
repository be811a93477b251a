// Fixed-base rows of the segmented small MSM (include/snarkv_amd.h `snarkv_g1_fixed_table_*`, `snarkv_g1_msm_batched_fixed`).
//
// Every proof of ONE protocol multiplies the same bases: its preprocessed commitments and the generator are 9 of the 21
// terms of `Gwc19::verify`'s left MSM (pcs/kzg/multiopen/gwc19.rs:124-139 over verifier/plonk/proof.rs:201-306), 9 of
// Bdfg21's 20 -- and `NativeLoader::multi_scalar_multiplication` (loader/native.rs:61-71) runs a 254-step double-and-add on
// each of them for every proof.  The reference's own in-circuit loader already splits fixed- from variable-base terms
// (loader/halo2/loader.rs:637-720).  Here a `snarkv_fixed_table` holds, per base B and 8-bit window j, the affine points
// d 2^(8 j) B for d = 1 .. 128 in HBM (33 windows x 128 entries x 64 B = 264 KiB per base: nine bases sit in one XCD's
// L2), built once per protocol BY THE VARIABLE-BASE KERNELS THEMSELVES (one-term segments of launch_msm_batched: the
// table is by construction what those kernels compute), and a fixed term s B becomes 33 table additions -- no doubling:
//   k_fixed_digits   lane per fixed term: the scalar (wire or in-memory form) recoded into signed 8-bit digits
//                    s = sum_j d_j 2^(8 j), d_j in [-128, 128]
//   k_fixed_terms    16 (latency) or 4 (throughput) lanes per SEGMENT walk its (term, window) pairs -- a branch-free mixed
//                    XYZZ addition each, the next table entry in flight --, a tree adds the lanes' sums, lane 0 stores
//                    the segment's fixed part (one XYZZ point); identity bases, duplicate bases and B next to -B are legal
//                    inputs: an exceptional case shows up as ZZ = 0 and the segment is redone with the careful adders
//   k_segment_fold   (msm_naive.hip) adds that point to the segment's variable-base partials before `to_affine`
// ~10 field products per (term, window) against ~1 200 - 3 600 per variable-base term; the kernels run on a side stream
// next to the variable-base ones.  Results are the bytes of snarkv_g1_msm_batched on the same terms (tests/test_gpu_fixed_base.py).
#include "ctx.hpp"
#include "g1_29.h"
#include "fr29.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

struct snarkv_fixed_table {
  int device;
  size_t n;        // bases
  void* d_table;   // G1Packed[n][kFixedWindows][kFixedEntries]
};

namespace snarkv {

constexpr uint32_t kFixedWindows = 33;   // 8-bit windows of a 256-bit integer + the carry out of the top one
constexpr uint32_t kFixedEntries = 128;  // |digit| = 1 .. 128


// canonical / in-memory words -> canonical affine words (the table is built in the wire form)
__global__ void k_fixed_bases_to_wire(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, uint32_t mont) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) w[j] = in[(size_t)i * 16 + j];
  G1Affine29 p = g1a29_from_words(w, mont != 0);
  g1a29_to_words(p, w, false);
#pragma unroll
  for (int j = 0; j < 16; ++j) out[(size_t)i * 16 + j] = w[j];
}

// term e = (base b, window j, digit d): scalar S[j][d] (the same for every base), point = base b; offsets[e] = e
__global__ void k_fixed_build_terms(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sc, uint32_t* __restrict__ scalars,
                                    uint32_t* __restrict__ points, uint32_t* __restrict__ offsets, uint32_t n_bases) {
  const uint32_t per = kFixedWindows * kFixedEntries;
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e > n_bases * per) return;
  offsets[e] = e;
  if (e == n_bases * per) return;
  const uint32_t b = e / per, jd = e % per;
#pragma unroll
  for (int k = 0; k < 8; ++k) scalars[(size_t)e * 8 + k] = sc[(size_t)jd * 8 + k];
#pragma unroll
  for (int k = 0; k < 16; ++k) points[(size_t)e * 16 + k] = bases[(size_t)b * 16 + k];
}

__global__ void k_fixed_pack(const uint32_t* __restrict__ affine_wire, G1Packed* __restrict__ table, uint32_t n) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  uint32_t w[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) w[j] = affine_wire[(size_t)e * 16 + j];
  table[e] = g1a29_pack(g1a29_from_words(w, false));  // the identity stays 64 zero bytes
}

// s = sum_j d_j 256^j with d_j in [-128, 128]: a byte v (+ carry) above 128 becomes v - 256 with a carry up.
// Stored per term as 33 magnitudes (bytes, in a 36-byte row) + the 32 sign bits (the carry digit of window 32 is 0 or +1).
__global__ void k_fixed_digits(const uint32_t* __restrict__ scalars, uint32_t n_fixed, uint8_t* __restrict__ mags,
                               uint32_t* __restrict__ signs, uint32_t mont) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_fixed) return;
  uint32_t k[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = scalars[(size_t)t * 8 + j];
  if (mont) {
    uint32_t c[8];
    fr_words_from_mont256(k, c);
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = c[j];
  }
  uint32_t carry = 0, s0 = 0;
  uint8_t* m = mags + (size_t)t * 36;
  for (uint32_t j = 0; j < 32; ++j) {
    uint32_t v = ((k[j >> 2] >> (8 * (j & 3))) & 0xFFu) + carry;
    carry = v > 128 ? 1u : 0u;
    uint32_t mag = carry ? 256u - v : v;
    m[j] = (uint8_t)mag;
    s0 |= carry << j;
  }
  m[32] = (uint8_t)carry;
  m[33] = m[34] = m[35] = 0;
  signs[t] = s0;
}

// One (term, window) pair of segment `seg`'s fixed list: the table entry its digit selects, negated for a negative digit;
// `none` for a zero digit, an id outside the table or an identity base (all-zero entry).
struct FixedPick {
  G1Packed e;
  bool neg, none;
};
__device__ __forceinline__ FixedPick fixed_pick(const G1Packed* __restrict__ table, const uint32_t* __restrict__ ids,
                                                const uint8_t* __restrict__ mags, const uint32_t* __restrict__ signs,
                                                uint32_t n_bases, uint32_t f0, uint32_t q) {
  FixedPick r;
  const uint32_t t = f0 + q / kFixedWindows, j = q % kFixedWindows;
  const uint32_t mag = mags[(size_t)t * 36 + j];
  const uint32_t b = ids[t];
  r.none = mag == 0 || b >= n_bases;  // (ids are checked on the host for host-pointer calls; out of range adds nothing)
  r.neg = j < 32 && ((signs[t] >> j) & 1u);
  if (!r.none) {
    r.e = table[((size_t)b * kFixedWindows + j) * kFixedEntries + (mag - 1)];
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) any |= r.e.w[k];
    r.none = any == 0;
  }
  return r;
}

// G lanes per SEGMENT (64 / G segments per wavefront) walk its (term, window) pairs with the branch-free mixed addition,
// the next table entry in flight while the current one is added; a G-wide tree adds the lanes' sums.  A lane or a tree
// step that met an exceptional case (an accumulator equal to +- the entry: repeated or opposite bases, crafted scalars)
// shows up as ZZ = 0 (sticky) -- the segment is then redone by ONE lane with the careful adders.
//   G = 16: latency (one job: a segment's 297 pairs are 19 additions + 4 tree levels per lane)
//   G = 4:  throughput (many jobs merged: 75 additions + 2 levels; the tree's idle lanes cost 6 % instead of 25 %)
template <uint32_t G>
__global__ void __launch_bounds__(64) k_fixed_terms(const G1Packed* __restrict__ table, const uint32_t* __restrict__ ids,
                                                     const uint32_t* __restrict__ foffs, const uint8_t* __restrict__ mags,
                                                     const uint32_t* __restrict__ signs, uint32_t n_msm, uint32_t n_bases,
                                                     G1Xyzz29* __restrict__ out) {
  __shared__ G1Xyzz29 sh[64];
  __shared__ uint32_t redo[64 / G];
  const uint32_t tid = threadIdx.x, lane = tid % G, sl = tid / G;
  const uint32_t seg = blockIdx.x * (64 / G) + sl;
  const bool live = seg < n_msm;
  const uint32_t f0 = live ? foffs[seg] : 0, f1 = live ? foffs[seg + 1] : 0;
  const uint32_t pairs = (f1 - f0) * kFixedWindows;
  if (lane == 0) redo[sl] = 0;
  G1Xyzz29 acc = xyzz29_identity();
  bool started = false;
  if (lane < pairs) {
    FixedPick cur = fixed_pick(table, ids, mags, signs, n_bases, f0, lane);
    for (uint32_t q = lane; q < pairs; q += G) {
      FixedPick nxt;
      nxt.none = true;
      if (q + G < pairs) nxt = fixed_pick(table, ids, mags, signs, n_bases, f0, q + G);  // in flight under the addition below
      if (!cur.none) {
        G1Affine29 p = g1a29_unpack(cur.e);
        if (cur.neg) p = g1a29_neg(p);
        if (started) {
          xyzz29_madd_fast(acc, p);
        } else {
          acc = xyzz29_from_affine(p);
          started = true;
        }
      }
      cur = nxt;
    }
  }
  bool bad = started && xyzz29_is_degenerate(acc);
  sh[tid] = acc;
  __syncthreads();
  for (uint32_t s = G / 2; s >= 1; s >>= 1) {
    if (lane < s) {
      G1Xyzz29 a = sh[tid];
      xyzz29_add_skipid_fast(a, sh[tid + s], bad);
      sh[tid] = a;
    }
    __syncthreads();
  }
  if (lane == 0) {
    const G1Xyzz29 r = sh[tid];
    bad = bad || (!xyzz29_is_identity(r) && xyzz29_is_degenerate(r));
  }
  if (bad) atomicOr(&redo[sl], 1u);
  __syncthreads();
  if (lane == 0 && live) {
    G1Xyzz29 r = sh[tid];
    if (redo[sl]) {  // rare: the whole segment again, carefully, on this lane
      r = xyzz29_identity();
      for (uint32_t q = 0; q < pairs; ++q) {
        FixedPick c = fixed_pick(table, ids, mags, signs, n_bases, f0, q);
        if (c.none) continue;
        G1Affine29 p = g1a29_unpack(c.e);
        if (c.neg) p = g1a29_neg(p);
        xyzz29_madd_careful(r, p);
      }
    }
    out[seg] = xyzz29_sanitize(r);
  }
}

// the scalars d 256^j mod r, d = 1 .. 128, j = 0 .. 32, canonical little-endian, [j][d - 1]
static void fixed_scalars(std::vector<uint8_t>& out) {
  static const uint32_t RL[8] = SNARKV_FR_R_LIMBS;
  uint64_t R[4];
  for (int i = 0; i < 4; ++i) R[i] = (uint64_t)RL[2 * i] | ((uint64_t)RL[2 * i + 1] << 32);
  auto geq = [&](const uint64_t a[5]) {
    if (a[4]) return true;
    for (int i = 3; i >= 0; --i)
      if (a[i] != R[i]) return a[i] > R[i];
    return true;
  };
  auto sub = [&](uint64_t a[5]) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 x = (unsigned __int128)a[i] - R[i] - (uint64_t)br;
      a[i] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
    a[4] -= (uint64_t)br;
  };
  auto add = [&](uint64_t a[5], const uint64_t b[5]) {  // a += b mod r; a, b < r
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (unsigned __int128)a[i] + b[i];
      a[i] = (uint64_t)c;
      c >>= 64;
    }
    a[4] = (uint64_t)c;
    if (geq(a)) sub(a);
  };
  out.assign((size_t)kFixedWindows * kFixedEntries * 32, 0);
  uint64_t base[5] = {1, 0, 0, 0, 0};  // 256^j mod r
  for (uint32_t j = 0; j < kFixedWindows; ++j) {
    uint64_t cur[5] = {0, 0, 0, 0, 0};
    for (uint32_t d = 1; d <= kFixedEntries; ++d) {
      add(cur, base);
      memcpy(&out[((size_t)j * kFixedEntries + (d - 1)) * 32], cur, 32);
    }
    for (int k = 0; k < 8; ++k) {  // base <- 256 base
      uint64_t t[5];
      memcpy(t, base, sizeof t);
      add(base, t);
    }
  }
}

int fixed_table_create(snarkv_ctx* ctx, const void* d_points_call_encoding, size_t n, snarkv_fixed_table** out) {
  const size_t per = (size_t)kFixedWindows * kFixedEntries, total = n * per;
  if (total >= ((size_t)1 << 31)) return SNARKV_ERR_LENGTH;
  snarkv_fixed_table* tab = new snarkv_fixed_table();
  tab->device = ctx->device;
  tab->n = n;
  tab->d_table = nullptr;
  auto fail = [&](int rc) {
    if (tab->d_table) (void)hipFree(tab->d_table);
    delete tab;
    return rc;
  };
  if (hipMalloc(&tab->d_table, total * sizeof(G1Packed)) != hipSuccess) {
    set_last_error("fixed_table_create: hipMalloc of %zu bytes failed", total * sizeof(G1Packed));
    return fail(SNARKV_ERR_DEVICE);
  }
  std::vector<uint8_t> sc;
  fixed_scalars(sc);
  void *d_sc = nullptr, *d_bases = nullptr, *d_s = nullptr, *d_p = nullptr, *d_o = nullptr, *d_aff = nullptr;
  int rc = SNARKV_OK;
  const uint32_t mont = ctx->mont ? 1u : 0u;
  // scratch of the build only: freed before returning (a one-off of a few MB per protocol)
  auto alloc = [&](void** p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; };
  if (!alloc(&d_sc, sc.size()) || !alloc(&d_bases, n * 64) || !alloc(&d_s, total * 32) || !alloc(&d_p, total * 64) ||
      !alloc(&d_o, (total + 1) * 4) || !alloc(&d_aff, total * 64)) {
    set_last_error("fixed_table_create: scratch allocation failed");
    rc = SNARKV_ERR_DEVICE;
  }
  if (rc == SNARKV_OK) {
    hipError_t e = hipMemcpyAsync(d_sc, sc.data(), sc.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) rc = SNARKV_ERR_DEVICE;
  }
  if (rc == SNARKV_OK) {
    hipLaunchKernelGGL(k_fixed_bases_to_wire, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, ctx->stream,
                       (const uint32_t*)d_points_call_encoding, (uint32_t*)d_bases, (uint32_t)n, mont);
    hipLaunchKernelGGL(k_fixed_build_terms, dim3((uint32_t)((total + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)d_bases, (const uint32_t*)d_sc, (uint32_t*)d_s, (uint32_t*)d_p, (uint32_t*)d_o, (uint32_t)n);
    if (hipGetLastError() != hipSuccess) rc = SNARKV_ERR_DEVICE;
  }
  if (rc == SNARKV_OK) {
    SNARKV_WIRE_FORM(ctx);  // the build runs in the wire form whatever the call's encoding
    rc = launch_msm_batched_ex(ctx, d_s, d_p, d_o, total, total, d_aff, nullptr, nullptr);
  }
  if (rc == SNARKV_OK) {
    hipLaunchKernelGGL(k_fixed_pack, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_aff,
                       (G1Packed*)tab->d_table, (uint32_t)total);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
      set_last_error("fixed_table_create: %s", hipGetErrorString(hipGetLastError()));
      rc = SNARKV_ERR_DEVICE;
    }
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  for (void* p : {d_sc, d_bases, d_s, d_p, d_o, d_aff})
    if (p) (void)hipFree(p);
  if (rc != SNARKV_OK) return fail(rc);
  *out = tab;
  return SNARKV_OK;
}

void fixed_table_free(snarkv_fixed_table* tab) {
  if (tab->d_table) (void)hipFree(tab->d_table);
  delete tab;
}
int fixed_table_device(const snarkv_fixed_table* tab) { return tab->device; }
size_t fixed_table_bases(const snarkv_fixed_table* tab) { return tab->n; }

// the fixed part of every segment -> d_extra[n_msm] (one XYZZ point each), enqueued on `st`
int launch_fixed_terms(snarkv_ctx* ctx, hipStream_t st, const snarkv_fixed_table* tab, const void* d_fixed_scalars,
                       const void* d_fixed_ids, const void* d_fixed_offsets, size_t n_msm, size_t n_fixed, void* d_mags,
                       void* d_signs, void* d_extra) {
  const uint32_t mont = ctx->mont ? 1u : 0u;
  if (n_fixed)
    hipLaunchKernelGGL(k_fixed_digits, dim3((uint32_t)((n_fixed + 63) / 64)), dim3(64), 0, st, (const uint32_t*)d_fixed_scalars,
                       (uint32_t)n_fixed, (uint8_t*)d_mags, (uint32_t*)d_signs, mont);
  // lanes per segment: 16 while the launch is a latency chain (a few thousand segments), 4 when it has to be cheap
  const char* eg = getenv("SNARKV_FIXED_LANES");  // A/B knob: 4 | 16
  const int g = eg ? atoi(eg) : (n_msm > 8192 || ctx->throughput_mode ? 4 : 16);
  if (g == 4)
    hipLaunchKernelGGL((k_fixed_terms<4>), dim3((uint32_t)((n_msm + 15) / 16)), dim3(64), 0, st, (const G1Packed*)tab->d_table,
                       (const uint32_t*)d_fixed_ids, (const uint32_t*)d_fixed_offsets, (const uint8_t*)d_mags,
                       (const uint32_t*)d_signs, (uint32_t)n_msm, (uint32_t)tab->n, (G1Xyzz29*)d_extra);
  else
    hipLaunchKernelGGL((k_fixed_terms<16>), dim3((uint32_t)((n_msm + 3) / 4)), dim3(64), 0, st, (const G1Packed*)tab->d_table,
                       (const uint32_t*)d_fixed_ids, (const uint32_t*)d_fixed_offsets, (const uint8_t*)d_mags,
                       (const uint32_t*)d_signs, (uint32_t)n_msm, (uint32_t)tab->n, (G1Xyzz29*)d_extra);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
