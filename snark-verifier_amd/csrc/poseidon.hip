// Poseidon transcripts of MANY proofs on the device.
//
// The reference's native PoseidonTranscript (snark-verifier/src/system/halo2/
// transcript/halo2.rs:170-321 over util/hash/poseidon.rs:115-202) hashes a few
// dozen Fr elements per proof: ~19 permutations of ~1 000 field products, half
// a millisecond of host time per proof -- more than everything else on the path
// once the EC work is on the GPU.  The hashing of one proof is sequential, but
// proofs are independent and the sequence of absorb / squeeze operations is
// fixed by the protocol, so a batch of transcripts is one kernel launch:
//
//   P0 k_poseidon_tables     : the optimised schedule's tables (exactly what the
//                              reference's `poseidon::Spec` holds) -> Montgomery
//                              form on the 9x29-bit scalar field (fr29.h)
//   P1 k_poseidon_transcript : 8 lanes per transcript, lane j = state word j;
//                              full round: x^5 + k in every lane, dense MDS row
//                              per lane (words exchanged by 8-lane shuffles);
//                              partial round: x^5 + k in lane 0, sparse matrix
//                              = one product per lane + a 3-step butterfly.
// Every transcript runs the same control flow (the segment lengths are shared),
// so there is no divergence.
#include <algorithm>
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "fr29.h"

struct snarkv_poseidon {
  int device;
  uint32_t t, rate, r_f, r_p;
  void* d_tables;  // Fr29 x n_tables, Montgomery
  uint32_t n_tables;
};

namespace snarkv {

// a*b + c*d, one reduction (inputs carry-normalised, |limb| < 2^29): the MDS dot products
__device__ __forceinline__ Fr29 fr29_mul2(const Fr29& a, const Fr29& b, const Fr29& c, const Fr29& d) {
  int32_t m[9];
  Fr29 r;
  int64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (int64_t)c.v[i] * d.v[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (int64_t)m[i] * fr29_r(k - i);
    m[k] = (int32_t)(((uint32_t)acc * (uint32_t)BN254_FR29_NINV) & (uint32_t)kMask29);
    acc += (int64_t)m[k] * fr29_r(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)c.v[i] * d.v[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (int64_t)m[i] * fr29_r(k - i);
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc >>= 29;
  }
  r.v[8] = (int32_t)acc;
  return r;
}

__global__ void k_poseidon_tables(const uint32_t* __restrict__ canon, Fr29* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = canon[(size_t)i * 8 + j];
  out[i] = fr29_canon_residue(fr29_from_canonical(w));
}

struct PoseidonShape {
  uint32_t t, rate, r_f, r_p;
  // offsets (in Fr29 elements) into the table
  uint32_t o_start, o_partial, o_end, o_mds, o_pre, o_rows, o_cols, n_tables;
};

// Exchanges inside the 8-lane group of one transcript, as DPP modifiers (no LDS crossbar):
//   broadcast of lane I: quad_perm [i,i,i,i] fills I's quad, a bank-masked row_shr:4 / row_shl:4
//   copies it into the other quad of the group (banks = quads of a 16-lane row);
//   sum over the group: xor-1 and xor-2 inside quads, then row_half_mirror swaps the two quads.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ int32_t dpp_keep(int32_t old, int32_t x) {
  return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xF, BANK_MASK, false);
}
template <int I>
__device__ __forceinline__ Fr29 bcast8(const Fr29& x) {
  constexpr int q = I & 3;
  constexpr int quad = q | (q << 2) | (q << 4) | (q << 6);
  Fr29 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    int32_t v = __builtin_amdgcn_update_dpp(0, x.v[k], quad, 0xF, 0xF, true);  // every quad: its own lane q
    // the quad that does NOT hold lane I takes the value from the one that does
    r.v[k] = I < 4 ? dpp_keep<0x114, 0xA>(v, v)   // row_shr:4 into the upper quads (banks 1, 3)
                   : dpp_keep<0x104, 0x5>(v, v);  // row_shl:4 into the lower quads (banks 0, 2)
  }
  return r;
}
__device__ __forceinline__ Fr29 shfl8(const Fr29& x, int src) {  // src is wavefront-uniform
  switch (src) {
    case 0: return bcast8<0>(x);
    case 1: return bcast8<1>(x);
    case 2: return bcast8<2>(x);
    case 3: return bcast8<3>(x);
    case 4: return bcast8<4>(x);
    case 5: return bcast8<5>(x);
    case 6: return bcast8<6>(x);
    default: return bcast8<7>(x);
  }
}
// every lane of the group gets the limb-wise sum over its 8 lanes (limbs as unsigned)
__device__ __forceinline__ Fr29 group8_sum_u(Fr29 x) {
#pragma unroll
  for (int k = 0; k < 9; ++k)
    x.v[k] = (int32_t)((uint32_t)x.v[k] + (uint32_t)__builtin_amdgcn_update_dpp(0, x.v[k], 0xB1, 0xF, 0xF, true));
#pragma unroll
  for (int k = 0; k < 9; ++k)
    x.v[k] = (int32_t)((uint32_t)x.v[k] + (uint32_t)__builtin_amdgcn_update_dpp(0, x.v[k], 0x4E, 0xF, 0xF, true));
#pragma unroll
  for (int k = 0; k < 9; ++k)
    x.v[k] = (int32_t)((uint32_t)x.v[k] + (uint32_t)__builtin_amdgcn_update_dpp(0, x.v[k], 0x141, 0xF, 0xF, true));
  return x;
}

// new word j = sum_i M[j][i] * s_i : pairs of products fused, lazy sum, one carry pass
__device__ __forceinline__ Fr29 dense_row(const Fr29* __restrict__ tab, uint32_t o_mat, uint32_t t, int j, const Fr29& s) {
  const int jj = j < (int)t ? j : 0;
  const Fr29* row = tab + o_mat + (uint32_t)jj * t;
  Fr29 acc = fr29_zero();
  uint32_t i = 0;
  for (; i + 1 < t; i += 2) {
    Fr29 s0 = shfl8(s, (int)i), s1 = shfl8(s, (int)i + 1);
    acc = fr29_add(acc, fr29_mul2(row[i], s0, row[i + 1], s1));
  }
  if (i < t) acc = fr29_add(acc, fr29_mul(row[i], shfl8(s, (int)i)));
  return fr29_norm(acc);  // <= 4 lazy terms of < 2^29 limbs for t <= 8
}

// 8 lanes per transcript; 32 transcripts per 256-thread block
__global__ void __launch_bounds__(256)
    k_poseidon_transcript(const Fr29* __restrict__ tables, PoseidonShape sh, const uint32_t* __restrict__ elems,
                          uint32_t n, uint32_t L, const uint32_t* __restrict__ seg_len, uint32_t S,
                          uint32_t* __restrict__ out) {
  extern __shared__ int32_t lds_raw[];
  Fr29* tab = reinterpret_cast<Fr29*>(lds_raw);
  for (uint32_t i = threadIdx.x; i < sh.n_tables; i += blockDim.x) tab[i] = tables[i];
  __syncthreads();
  const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;  // transcript
  const int j = threadIdx.x & 7;                                     // state word
  const bool live = g < n;
  const uint32_t gi = live ? g : 0;
  const uint32_t t = sh.t, h = sh.r_f / 2;
  // poseidon::State::default(): word 0 = 2^64
  Fr29 s = fr29_zero();
  if (j == 0) {
    uint32_t w[8] = {0, 0, 1, 0, 0, 0, 0, 0};
    s = fr29_canon_residue(fr29_from_canonical(w));
  }
  const Fr29 one = fr29_one();

  auto permutation = [&](uint32_t first, uint32_t cnt) {
    // absorb (poseidon.rs:44-75): inputs into words 1.., a one after the last input, the pre-constants
    if (j >= 1 && (uint32_t)j <= cnt) {
      uint32_t w[8];
      const uint32_t* src = elems + ((size_t)gi * L + first + (uint32_t)j - 1) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) w[k] = src[k];
      s = fr29_add(s, fr29_from_canonical(w));
    }
    if ((uint32_t)j == cnt + 1 && cnt + 1 < t) s = fr29_add(s, one);
    if ((uint32_t)j < t) s = fr29_norm(fr29_add(s, tab[sh.o_start + (uint32_t)j]));
    // first half of the full rounds; the last one is followed by the pre-sparse matrix
    for (uint32_t r = 0; r < h; ++r) {
      Fr29 x = fr29_pow5(s);
      if ((uint32_t)j < t) x = fr29_norm(fr29_add(x, tab[sh.o_start + (r + 1) * t + (uint32_t)j]));
      s = dense_row(tab, r + 1 < h ? sh.o_mds : sh.o_pre, t, j, x);
    }
    // partial rounds: S-box on word 0 only, sparse matrix [[row], [col_hat | I]]
    // The three table entries of a round (row[j], col_hat[j-1] / row[0], the round constant) are read one round AHEAD:
    // the LDS latency of round r + 1's operands passes under round r's products instead of in front of its first one.
    // (lanes >= t read row[0] / compute products nobody uses: one table read by index, no per-lane struct selection)
    const uint32_t jr = (uint32_t)j < t ? (uint32_t)j : 0u;
    const bool has_col = j >= 1 && (uint32_t)j < t;
    auto col_index = [&](uint32_t r) { return has_col ? sh.o_cols + r * (t - 1) + (uint32_t)j - 1 : sh.o_rows + r * t; };
    Fr29 rowj_next = tab[sh.o_rows + jr], cj_next = tab[col_index(0)], kc_next = tab[sh.o_partial];
    for (uint32_t r = 0; r < sh.r_p; ++r) {
      // Lanes run in lockstep, so the three products of lane 0's x^5 are issued by every lane anyway:
      // the first one doubles as  row[j] * s_j  in lanes 1.. (operands picked per lane).
      const bool w0 = j == 0;
      const Fr29 rowj = rowj_next, cj = cj_next, kc = kc_next;
      {
        const uint32_t rn = r + 1 < sh.r_p ? r + 1 : r;
        rowj_next = tab[sh.o_rows + rn * t + jr];
        cj_next = tab[col_index(rn)];
        kc_next = tab[sh.o_partial + rn];
      }
      Fr29 p1 = fr29_mul(w0 ? s : rowj, s);  // lane 0: x^2      lanes 1..: row[j] * s_j
      Fr29 x4 = fr29_mul(p1, p1);            // lane 0: x^4
      Fr29 x5 = fr29_mul(x4, s);             // lane 0: x^5
      Fr29 s0 = fr29_norm(fr29_add(x5, kc));
      s0 = shfl8(s0, 0);
      // second shared product: lane 0: row[0] * s0      lanes 1..: col_hat[j-1] * s0
      Fr29 p2 = fr29_mul(cj, s0);  // lane 0: cj = row[0]
      // word 0: row . state  (butterfly over the 8 lanes; lanes >= t contribute zero)
      Fr29 sum = w0 ? p2 : p1;
      if ((uint32_t)j >= t) sum = fr29_zero();
      sum = group8_sum_u(sum);
      // limbs 0..7 of `sum` are sums of <= 8 values in [0, 2^29): read them as unsigned
      Fr29 n0;
      {
        int64_t c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          int64_t v = (int64_t)(uint32_t)sum.v[k] + c;
          n0.v[k] = (int32_t)v & kMask29;
          c = v >> 29;
        }
        n0.v[8] = (int32_t)((int64_t)sum.v[8] + c);
      }
      // words 1..: s_j + col_hat[j-1] * s0
      Fr29 nj = s;
      if (j >= 1 && (uint32_t)j < t) nj = fr29_norm(fr29_add(s, p2));
      s = w0 ? n0 : nj;
    }
    // second half of the full rounds; no constant after the last S-box
    for (uint32_t r = 0; r < h; ++r) {
      Fr29 x = fr29_pow5(s);
      if (r + 1 < h && (uint32_t)j < t) x = fr29_norm(fr29_add(x, tab[sh.o_end + r * t + (uint32_t)j]));
      s = dense_row(tab, sh.o_mds, t, j, x);
    }
  };

  uint32_t pos = 0;
  for (uint32_t q = 0; q < S; ++q) {  // `squeeze` (poseidon.rs:151-164)
    uint32_t len = seg_len[q];
    const bool exact = len % sh.rate == 0;
    for (uint32_t off = 0; off < len; off += sh.rate) permutation(pos + off, len - off < sh.rate ? len - off : sh.rate);
    if (exact) permutation(pos + len, 0);
    pos += len;
    Fr29 o = shfl8(s, 1);  // the challenge is state word 1
    if (live && j == 0) {
      uint32_t w[8];
      fr29_to_canonical(o, w);
      uint32_t* dst = out + ((size_t)g * S + q) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) dst[k] = w[k];
    }
  }
}

// ---- proof-shaped transcripts: the absorbed elements assembled on the device ------------------------------------------
// What a native PoseidonTranscript absorbs while reading a proof (halo2.rs:215-275) is, element by element, one of: a
// value the caller brought (initial state, instances: LEAD), a scalar of the proof (32 bytes at a fixed offset), or a
// coordinate of a point of the proof reduced mod r (`fe_to_fe`; p < 2r: one conditional subtraction).  The order is the
// protocol's, so a batch needs it once (`layout`), and the device can build every transcript's input from the proofs
// as they are and the points it has just decompressed -- no host pass over the batch before the hashing.
constexpr uint32_t kPsrcLead = 0u, kPsrcScalar = 1u, kPsrcPx = 2u, kPsrcPy = 3u;  // layout code = kind << 28 | value

__global__ void __launch_bounds__(256)
    k_poseidon_gather(const uint32_t* __restrict__ proofs, uint32_t stride_words, const uint32_t* __restrict__ lead, uint32_t n_lead,
                      const uint32_t* __restrict__ layout, uint32_t L, const uint32_t* __restrict__ pts, uint32_t P, uint32_t n,
                      uint32_t* __restrict__ elems) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n * L) return;
  const uint32_t i = id / L, k = id % L, code = layout[k], kind = code >> 28, v = code & 0x0FFFFFFFu;
  const uint32_t* src = kind == kPsrcLead     ? lead + ((size_t)i * n_lead + v) * 8
                        : kind == kPsrcScalar ? proofs + (size_t)i * stride_words + (v >> 2)
                                              : pts + ((size_t)i * P + v) * 16 + (kind == kPsrcPy ? 8 : 0);
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = src[j];
  if (kind >= kPsrcPx) {  // a coordinate (< p < 2r) as an element of Fr
    constexpr uint32_t r[8] = SNARKV_FR_R_LIMBS;
    uint32_t d[8], borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint64_t x = (uint64_t)w[j] - r[j] - borrow;
      d[j] = (uint32_t)x;
      borrow = (uint32_t)(x >> 63);
    }
    if (!borrow) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = d[j];
    }
  }
  uint32_t* dst = elems + (size_t)id * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = w[j];
}

int launch_g1_decompress_records(snarkv_ctx* ctx, const void* d_records, size_t n_rec, size_t stride_words, const void* d_offs_words,
                                 size_t P, void* d_out64, void* d_ok);

}  // namespace snarkv

using namespace snarkv;

extern "C" {

int snarkv_poseidon_create(snarkv_ctx* ctx, uint32_t t, uint32_t rate, uint32_t r_f, uint32_t r_p, const uint8_t* start,
                           const uint8_t* partial, const uint8_t* end, const uint8_t* mds,
                           const uint8_t* pre_sparse_mds, const uint8_t* sparse_rows, const uint8_t* sparse_col_hats,
                           snarkv_poseidon** out) {
  if (!ctx || !start || !partial || !end || !mds || !pre_sparse_mds || !sparse_rows || !sparse_col_hats || !out)
    return SNARKV_ERR_ARG;
  if (t < 2 || t > 8 || rate == 0 || rate >= t || r_f < 2 || (r_f & 1) || r_p == 0 || r_p > 4096) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipSetDevice(ctx->device));
  const uint32_t h = r_f / 2;
  const uint32_t n_start = (h + 1) * t, n_end = (h - 1) * t, n_rows = r_p * t, n_cols = r_p * (t - 1);
  const uint32_t total = n_start + r_p + n_end + 2 * t * t + n_rows + n_cols;
  if ((size_t)total * sizeof(Fr29) > 60 * 1024) return SNARKV_ERR_ARG;  // tables must fit LDS
  std::vector<uint8_t> host;
  host.reserve((size_t)total * 32);
  host.insert(host.end(), start, start + (size_t)n_start * 32);
  host.insert(host.end(), partial, partial + (size_t)r_p * 32);
  host.insert(host.end(), end, end + (size_t)n_end * 32);
  host.insert(host.end(), mds, mds + (size_t)t * t * 32);
  host.insert(host.end(), pre_sparse_mds, pre_sparse_mds + (size_t)t * t * 32);
  host.insert(host.end(), sparse_rows, sparse_rows + (size_t)n_rows * 32);
  host.insert(host.end(), sparse_col_hats, sparse_col_hats + (size_t)n_cols * 32);
  snarkv_poseidon* ps = new snarkv_poseidon();
  ps->device = ctx->device;
  ps->t = t;
  ps->rate = rate;
  ps->r_f = r_f;
  ps->r_p = r_p;
  ps->n_tables = total;
  ps->d_tables = nullptr;
  void* d_canon = nullptr;
  if (hipMalloc(&ps->d_tables, (size_t)total * sizeof(Fr29)) != hipSuccess ||
      hipMalloc(&d_canon, host.size()) != hipSuccess) {
    if (ps->d_tables) (void)hipFree(ps->d_tables);
    delete ps;
    return SNARKV_ERR_DEVICE;
  }
  SNARKV_HIP(hipMemcpyAsync(d_canon, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_poseidon_tables, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, (const uint32_t*)d_canon,
                     (Fr29*)ps->d_tables, total);
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  (void)hipFree(d_canon);
  *out = ps;
  return SNARKV_OK;
}

void snarkv_poseidon_destroy(snarkv_poseidon* ps) {
  if (!ps) return;
  (void)hipSetDevice(ps->device);
  if (ps->d_tables) (void)hipFree(ps->d_tables);
  delete ps;
}

static PoseidonShape shape_of(const snarkv_poseidon* ps) {
  PoseidonShape sh;
  const uint32_t t = ps->t, h = ps->r_f / 2;
  sh.t = t;
  sh.rate = ps->rate;
  sh.r_f = ps->r_f;
  sh.r_p = ps->r_p;
  sh.o_start = 0;
  sh.o_partial = sh.o_start + (h + 1) * t;
  sh.o_end = sh.o_partial + ps->r_p;
  sh.o_mds = sh.o_end + (h - 1) * t;
  sh.o_pre = sh.o_mds + t * t;
  sh.o_rows = sh.o_pre + t * t;
  sh.o_cols = sh.o_rows + ps->r_p * t;
  sh.n_tables = ps->n_tables;
  return sh;
}

int snarkv_poseidon_transcript_batch_dev(snarkv_ctx* ctx, const snarkv_poseidon* ps, const void* d_elems, size_t n,
                                         size_t L, const void* d_seg_len, size_t S, void* d_out) {
  if (!ctx || !ps || !d_seg_len || !d_out || (!d_elems && L)) return SNARKV_ERR_ARG;
  if (n == 0 || S == 0) return SNARKV_ERR_EMPTY;
  if (n >= ((size_t)1 << 28) || L >= ((size_t)1 << 24)) return SNARKV_ERR_LENGTH;
  SNARKV_HIP(hipSetDevice(ctx->device));
  PoseidonShape sh = shape_of(ps);
  uint32_t blocks = (uint32_t)((n * 8 + 255) / 256);
  hipLaunchKernelGGL(k_poseidon_transcript, dim3(blocks), dim3(256), (size_t)sh.n_tables * sizeof(Fr29), ctx->stream,
                     (const Fr29*)ps->d_tables, sh, (const uint32_t*)d_elems, (uint32_t)n, (uint32_t)L,
                     (const uint32_t*)d_seg_len, (uint32_t)S, (uint32_t*)d_out);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

int snarkv_poseidon_transcript_batch(snarkv_ctx* ctx, const snarkv_poseidon* ps, const uint8_t* elems, size_t n, size_t L,
                                     const uint32_t* seg_len, size_t S, uint8_t* out) {
  if (!ctx || !ps || !seg_len || !out || (!elems && L)) return SNARKV_ERR_ARG;
  if (n == 0 || S == 0) return SNARKV_ERR_EMPTY;
  size_t sum = 0;
  for (size_t q = 0; q < S; ++q) sum += seg_len[q];
  if (sum != L) return SNARKV_ERR_LENGTH;
  SNARKV_HIP(hipSetDevice(ctx->device));
  void *d_e = nullptr, *d_s = nullptr, *d_o = nullptr;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_SCALARS, std::max<size_t>(32, n * L * 32), &d_e));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_OFFSETS, S * 4, &d_s));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, n * S * 32, &d_o));
  if (n * L) SNARKV_HIP(hipMemcpyAsync(d_e, elems, n * L * 32, hipMemcpyHostToDevice, ctx->stream));
  SNARKV_HIP(hipMemcpyAsync(d_s, seg_len, S * 4, hipMemcpyHostToDevice, ctx->stream));
  SNARKV_TRY(snarkv_poseidon_transcript_batch_dev(ctx, ps, d_e, n, L, d_s, S, d_o));
  SNARKV_HIP(hipMemcpyAsync(out, d_o, n * S * 32, hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  return SNARKV_OK;
}

int snarkv_poseidon_read_batch(snarkv_ctx* ctx, const snarkv_poseidon* ps, const uint8_t* proofs, size_t n, size_t stride,
                               const uint8_t* lead, size_t n_lead, const uint32_t* layout, size_t L,
                               const uint32_t* point_offsets, size_t P, const uint32_t* seg_len, size_t S, uint8_t* challenges,
                               uint8_t* points64, uint8_t* ok) {
  if (!ctx || !ps || !proofs || !layout || !seg_len || !challenges || (n_lead && !lead) || (P && (!point_offsets || !points64 || !ok)))
    return SNARKV_ERR_ARG;
  if (n == 0 || S == 0 || L == 0) return SNARKV_ERR_EMPTY;
  if (n >= ((size_t)1 << 24) || L >= ((size_t)1 << 16) || stride >= ((size_t)1 << 28) || (stride & 15) || P >= ((size_t)1 << 16))
    return SNARKV_ERR_LENGTH;
  // the gather and decompress kernels index elements / points with 32 bits
  if (n * L >= ((size_t)1 << 32) || n * P >= ((size_t)1 << 32)) return SNARKV_ERR_LENGTH;
  size_t sum = 0;
  for (size_t q = 0; q < S; ++q) sum += seg_len[q];
  if (sum != L) return SNARKV_ERR_LENGTH;
  std::vector<uint32_t> head(L + P + S);  // layout | point offsets in words | segment lengths: one small upload
  for (size_t k = 0; k < L; ++k) {
    const uint32_t kind = layout[k] >> 28, v = layout[k] & 0x0FFFFFFFu;
    if (kind > kPsrcPy) return SNARKV_ERR_ARG;
    if (kind == kPsrcLead && v >= n_lead) return SNARKV_ERR_ARG;
    if (kind == kPsrcScalar && ((v & 3) || (size_t)v + 32 > stride)) return SNARKV_ERR_ARG;
    if (kind >= kPsrcPx && v >= P) return SNARKV_ERR_ARG;
    head[k] = layout[k];
  }
  for (size_t q = 0; q < P; ++q) {
    if ((point_offsets[q] & 15) || (size_t)point_offsets[q] + 32 > stride) return SNARKV_ERR_ARG;
    head[L + q] = point_offsets[q] >> 2;
  }
  for (size_t q = 0; q < S; ++q) head[L + P + q] = seg_len[q];
  SNARKV_HIP(hipSetDevice(ctx->device));
  void *d_proofs = nullptr, *d_lead = nullptr, *d_head = nullptr, *d_pts = nullptr, *d_elems = nullptr, *d_out = nullptr;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_POINTS, n * stride, &d_proofs));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_POINTS_MONT, std::max<size_t>(32, n * n_lead * 32), &d_lead));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_OFFSETS, head.size() * 4, &d_head));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_TERM_PARTIALS, std::max<size_t>(80, n * P * 65), &d_pts));  // points, then a flag each
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_SCALARS, n * L * 32, &d_elems));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, n * S * 32, &d_out));
  SNARKV_HIP(hipMemcpyAsync(d_proofs, proofs, n * stride, hipMemcpyHostToDevice, ctx->stream));
  if (n_lead) SNARKV_HIP(hipMemcpyAsync(d_lead, lead, n * n_lead * 32, hipMemcpyHostToDevice, ctx->stream));
  SNARKV_HIP(hipMemcpyAsync(d_head, head.data(), head.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  const uint32_t* dh = (const uint32_t*)d_head;
  uint8_t* d_ok = (uint8_t*)d_pts + n * P * 64;
  if (P) SNARKV_TRY(launch_g1_decompress_records(ctx, d_proofs, n, stride / 4, dh + L, P, d_pts, d_ok));
  hipLaunchKernelGGL(k_poseidon_gather, dim3((uint32_t)((n * L + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_proofs,
                     (uint32_t)(stride / 4), (const uint32_t*)d_lead, (uint32_t)n_lead, dh, (uint32_t)L, (const uint32_t*)d_pts,
                     (uint32_t)P, (uint32_t)n, (uint32_t*)d_elems);
  SNARKV_HIP(hipGetLastError());
  if (P) {  // the points go home under the hashing (the context's copy stream, ordered behind the gather by an event)
    if (!ctx->copy_ready) {
      SNARKV_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
      ctx->copy_ready = true;
    }
    if (!ctx->sorted_ev_ready) {
      SNARKV_HIP(hipEventCreateWithFlags(&ctx->sorted_ev, hipEventDisableTiming));
      ctx->sorted_ev_ready = true;
    }
    SNARKV_HIP(hipEventRecord(ctx->sorted_ev, ctx->stream));
    SNARKV_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->sorted_ev, 0));
    SNARKV_HIP(hipMemcpyAsync(points64, d_pts, n * P * 64, hipMemcpyDeviceToHost, ctx->copy_stream));
    SNARKV_HIP(hipMemcpyAsync(ok, d_ok, n * P, hipMemcpyDeviceToHost, ctx->copy_stream));
  }
  int rc = snarkv_poseidon_transcript_batch_dev(ctx, ps, d_elems, n, L, dh + L + P, S, d_out);
  hipError_t e1 = rc == SNARKV_OK ? hipMemcpyAsync(challenges, d_out, n * S * 32, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
  // success or not, nothing queued above may still write the caller's buffers after the return
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  hipError_t e3 = P ? hipStreamSynchronize(ctx->copy_stream) : hipSuccess;
  if (rc != SNARKV_OK) return rc;
  SNARKV_HIP(e1);
  SNARKV_HIP(e2);
  SNARKV_HIP(e3);
  return SNARKV_OK;
}

}  // extern "C"
