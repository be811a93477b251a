// `IpaAs::decide` on the device (reference snark-verifier/src/pcs/ipa/decider.rs:47-55):
//     U == multi_scalar_multiplication(h_coeffs(xi, 1), dk.g).to_affine()
// The committing key G (2^k points) is uploaded ONCE into a deciding-key object -- the
// reference re-reads it from memory on every decide, a PCIe copy of 64 * 2^k bytes here --
// and the 2^k coefficients of h(X) = prod_i (1 + xi_{k-1-i} X^(2^i)) (pcs/ipa.rs:405-421)
// are produced by a kernel straight into the scalar buffer of the Pippenger, so a decide
// moves k scalars in and 64 bytes out.
#include <string.h>
#include <vector>
#include "ctx.hpp"
#include "fr29.h"

struct snarkv_ipa_dk {
  int device;
  uint32_t k;
  void* d_points;  // the points held: 64 B canonical affine each, as the Pippenger entry point takes them
  size_t first;    // index of the first point held in the 2^k-point key (0 unless a multi-GPU shard)
  size_t count;    // points held (2^k unless a shard)
};

namespace snarkv {

// coeff[j] = prod over the set bits i of j of xi[k-1-i]   (h_coeffs with scalar = 1: the
// doubling loop `coeffs[len + j] = coeffs[j] * xi` unrolled per index).  One lane per
// coefficient, <= k products; canonical little-endian out.
__global__ void __launch_bounds__(256)
    k_h_coeffs(const uint32_t* __restrict__ xi_canon, uint32_t k, uint32_t first, uint32_t count,
               uint32_t* __restrict__ out) {
  __shared__ Fr29 sx[32];
  if (threadIdx.x < k) sx[threadIdx.x] = fr29_from_canonical(xi_canon + 8 * (size_t)(k - 1 - threadIdx.x));
  __syncthreads();
  uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= count) return;
  const uint32_t j = first + t;  // coefficient index in the full 2^k vector; out[] is shard-local
  Fr29 acc = fr29_one();
#pragma unroll 1
  for (uint32_t i = 0; i < k; ++i)
    if ((j >> i) & 1u) acc = fr29_mul(acc, sx[i]);
  uint32_t w[8];
  fr29_to_canonical(acc, w);
  uint4* o = reinterpret_cast<uint4*>(out + 8 * (size_t)t);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

}  // namespace snarkv

using namespace snarkv;

extern "C" {

static int ipa_dk_make(snarkv_ctx* ctx, const uint8_t* g_points64, uint32_t k, size_t first, size_t count,
                       snarkv_ipa_dk** out) {
  SNARKV_HIP(hipSetDevice(ctx->device));
  snarkv_ipa_dk* dk = new snarkv_ipa_dk();
  dk->device = ctx->device;
  dk->k = k;
  dk->first = first;
  dk->count = count;
  dk->d_points = nullptr;
  if (hipMalloc(&dk->d_points, count * 64) != hipSuccess) {
    delete dk;
    set_last_error("ipa_dk_create: hipMalloc of %zu bytes failed", count * 64);
    return SNARKV_ERR_DEVICE;
  }
  SNARKV_HIP(hipMemcpyAsync(dk->d_points, g_points64, count * 64, hipMemcpyHostToDevice, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  *out = dk;
  return SNARKV_OK;
}

int SNARKV_API(ipa_dk_create)(snarkv_ctx* ctx, const uint8_t* g_points64, size_t n, snarkv_ipa_dk** out) {
  if (!ctx || !g_points64 || !out) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_ERR_EMPTY;
  uint32_t k = 0;
  while (((size_t)1 << k) < n) ++k;
  if (((size_t)1 << k) != n || k < 1 || k > 28) return SNARKV_ERR_LENGTH;  // committing keys have 2^k points
  return ipa_dk_make(ctx, g_points64, k, 0, n, out);
}

// Multi-GPU: this rank holds points [first, first + count) of the 2^k-point key.
int SNARKV_API(ipa_dk_create_shard)(snarkv_ctx* ctx, const uint8_t* g_shard64, size_t count, uint32_t k, size_t first,
                                    snarkv_ipa_dk** out) {
  if (!ctx || !g_shard64 || !out) return SNARKV_ERR_ARG;
  if (count == 0) return SNARKV_ERR_EMPTY;
  if (k < 1 || k > 28 || first + count > ((size_t)1 << k)) return SNARKV_ERR_LENGTH;
  return ipa_dk_make(ctx, g_shard64, k, first, count, out);
}

// The shard's part of commit(G, h(xi)) as a projective partial (SNARKV_G1_PARTIAL_BYTES), device to device:
// all-gather the partials and fold them (snarkv_g1_fold_partials_dev) to get the point `decide` compares with U.
int SNARKV_API(ipa_commit_partial_dev)(snarkv_ctx* ctx, const snarkv_ipa_dk* dk, const uint8_t* xi32, void* d_partial) {
  if (!ctx || !dk || !xi32 || !d_partial) return SNARKV_ERR_ARG;
  if (dk->device != ctx->device) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipSetDevice(ctx->device));
  SNARKV_WIRE_FORM(ctx);  // the IPA speaks the wire form whatever the context's default flags say (include/snarkv_amd.h)
  const uint32_t k = dk->k;
  void *d_xi, *d_h;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IPA_XI, (size_t)k * 32, &d_xi));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IPA_H, dk->count * 32, &d_h));
  SNARKV_HIP(hipMemcpyAsync(d_xi, xi32, (size_t)k * 32, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_h_coeffs, dim3((uint32_t)((dk->count + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint32_t*)d_xi, k, (uint32_t)dk->first, (uint32_t)dk->count, (uint32_t*)d_h);
  SNARKV_HIP(hipGetLastError());
  return launch_msm_pippenger(ctx, d_h, dk->d_points, dk->count, 0, d_partial, true);
}

void SNARKV_API(ipa_dk_destroy)(snarkv_ipa_dk* dk) {
  if (!dk) return;
  (void)hipSetDevice(dk->device);
  if (dk->d_points) (void)hipFree(dk->d_points);
  delete dk;
}

uint32_t SNARKV_API(ipa_dk_k)(const snarkv_ipa_dk* dk) { return dk ? dk->k : 0; }

int SNARKV_API(ipa_decide_batch)(snarkv_ctx* ctx, const snarkv_ipa_dk* dk, const uint8_t* xi32, const uint8_t* u64, size_t m,
                            uint8_t* ok) {
  if (!ctx || !dk || !xi32 || !u64 || !ok) return SNARKV_ERR_ARG;
  if (m == 0) return SNARKV_ERR_EMPTY;
  if (dk->device != ctx->device) return SNARKV_ERR_ARG;
  if (dk->first != 0 || dk->count != ((size_t)1 << dk->k)) return SNARKV_ERR_LENGTH;  // a shard cannot decide alone
  SNARKV_HIP(hipSetDevice(ctx->device));
  SNARKV_WIRE_FORM(ctx);  // (the lanes take the call's encoding at the fork: ctx_lanes_fork)
  const uint32_t k = dk->k;
  const size_t n = (size_t)1 << k;
  void *d_xi, *d_out;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IPA_XI, m * k * 32, &d_xi));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IPA_OUT, m * 64, &d_out));
  SNARKV_HIP(hipMemcpyAsync(d_xi, xi32, m * k * 32, hipMemcpyHostToDevice, ctx->stream));
  // accumulators are independent: up to four in flight (the tail of one Pippenger -- bucket reduce,
  // shift chains -- overlaps the accumulation of the next), each lane with its own h buffer
  const bool lanes = m >= 2;
  if (lanes) {
    SNARKV_TRY(ctx_lanes(ctx));
    SNARKV_TRY(ctx_lanes_fork(ctx));
  }
  // an error between fork and join must not leave the sub-streams unjoined: remember it, join, then return it
  int rc = SNARKV_OK;
  const bool was_throughput = ctx->throughput_mode;
  if (lanes) ctx->throughput_mode = true;  // four accumulators in flight: the caller's lane runs like the others
  for (size_t a = 0; a < m && rc == SNARKV_OK; ++a) {
    snarkv_ctx* lane = lanes ? ctx_lane(ctx, a) : ctx;
    void* d_h;
    rc = ctx_reserve(lane, SLOT_IPA_H, n * 32, &d_h);
    if (rc != SNARKV_OK) break;
    hipLaunchKernelGGL(k_h_coeffs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, lane->stream,
                       (const uint32_t*)d_xi + a * k * 8, k, 0u, (uint32_t)n, (uint32_t*)d_h);
    if (hipGetLastError() != hipSuccess) {
      set_last_error("ipa_decide_batch: k_h_coeffs launch failed");
      rc = SNARKV_ERR_DEVICE;
      break;
    }
    rc = launch_msm_pippenger(lane, d_h, dk->d_points, n, 0, (uint8_t*)d_out + 64 * a, false);
  }
  ctx->throughput_mode = was_throughput;
  if (lanes) {
    int jrc = ctx_lanes_join(ctx);
    if (rc == SNARKV_OK) rc = jrc;
  }
  if (rc != SNARKV_OK) return rc;
  std::vector<uint8_t> got(m * 64);
  SNARKV_HIP(hipMemcpyAsync(got.data(), d_out, m * 64, hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t a = 0; a < m; ++a) ok[a] = memcmp(&got[64 * a], u64 + 64 * a, 64) == 0 ? 1 : 0;
  return SNARKV_OK;
}

}  // extern "C"
