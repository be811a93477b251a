// KZG pairing decider on the device.
// Replaces the native `AccumulationDecider` impl for `KzgAs`
// (reference snark-verifier/src/pcs/kzg/decider.rs:70-93):
//   decide     : e(lhs, g2) * e(rhs, -s_g2) == 1
//   decide_all : the same for every accumulator of a list (decider.rs:84-93)
//
//   D0 k_g2_prepare : once per deciding key -- line tables of g2 and -s_g2
//                     (the reference's `G2Prepared::from`, redone there on
//                     every call, decider.rs:74)
//   D1 k_decide     : one lane per accumulator: 2-pair Miller loop with shared
//                     squarings + final exponentiation + `is_identity`.
// Batches of independent accumulators are the parallel axis;
// a single decide is latency-bound by construction.
#include "ctx.hpp"
#include "g1.cuh"
#include "pairing.cuh"

namespace snarkv {

size_t g2_prepared_bytes() { return sizeof(G2Prepared); }

__device__ __forceinline__ Fq load_fq_canonical(const uint32_t* __restrict__ src) {
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = src[i];
  return fq_from_canonical(w);
}

__device__ __forceinline__ void store_fq_canonical(const Fq& a, uint32_t* __restrict__ dst) {
  uint32_t w[8];
  fq_to_canonical(a, w);
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = w[i];
}

// g2x2: g2 (128 B canonical) || s_g2 (128 B canonical).  Lane 0 prepares g2,
// lane 1 prepares -s_g2.
__global__ void __launch_bounds__(64) k_g2_prepare(const uint32_t* __restrict__ g2x2, G2Prepared* __restrict__ prep) {
  uint32_t k = threadIdx.x;
  if (k >= 2) return;
  const uint32_t* src = g2x2 + k * 32;
  G2Affine q;
  q.x.c0 = load_fq_canonical(src);
  q.x.c1 = load_fq_canonical(src + 8);
  q.y.c0 = load_fq_canonical(src + 16);
  q.y.c1 = load_fq_canonical(src + 24);
  if (k == 1) q.y = fq2_neg(q.y);
  g2_prepare(q, prep[k]);
}

// y^2 == x^3 + 3/(9+u) and canonical coordinates, for both G2 points.
__global__ void __launch_bounds__(64) k_validate_g2(const uint32_t* __restrict__ g2x2, int* __restrict__ bad) {
  uint32_t k = threadIdx.x;
  if (k >= 2) return;
  const uint32_t* src = g2x2 + k * 32;
  bool ok = true;
  for (int j = 0; j < 4; ++j) ok = ok && fq_canonical_in_range(src + 8 * j);
  if (ok) {
    G2Affine q;
    q.x.c0 = load_fq_canonical(src);
    q.x.c1 = load_fq_canonical(src + 8);
    q.y.c0 = load_fq_canonical(src + 16);
    q.y.c1 = load_fq_canonical(src + 24);
    if (!(fq2_is_zero(q.x) && fq2_is_zero(q.y))) {
      constexpr uint32_t bc0[8] = BN254_TWIST_B_C0_MONT;
      constexpr uint32_t bc1[8] = BN254_TWIST_B_C1_MONT;
      Fq2 b;
      for (int i = 0; i < 8; ++i) {
        b.c0.v[i] = bc0[i];
        b.c1.v[i] = bc1[i];
      }
      Fq2 lhs = fq2_sqr(q.y);
      Fq2 rhs = fq2_add(fq2_mul(fq2_sqr(q.x), q.x), b);
      ok = fq2_eq(lhs, rhs);
    }
  }
  if (!ok) atomicAdd(bad, 1);
}

__global__ void __launch_bounds__(64)
    k_decide(const G2Prepared* __restrict__ prep, const uint32_t* __restrict__ accs, uint32_t m,
             uint8_t* __restrict__ ok, uint32_t* __restrict__ gt_out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint32_t* a = accs + (size_t)i * 32;
  G1AffineM ps[2];
  ps[0].x = load_fq_canonical(a);
  ps[0].y = load_fq_canonical(a + 8);
  ps[1].x = load_fq_canonical(a + 16);
  ps[1].y = load_fq_canonical(a + 24);
  const G2Prepared* qs[2] = {&prep[0], &prep[1]};
  Fq12 f = multi_miller_loop(ps, qs, 2);
  Fq12 e = final_exponentiation(f);
  if (ok) ok[i] = fq12_is_one(e) ? 1 : 0;
  if (gt_out) {
    uint32_t* g = gt_out + (size_t)i * 96;
    const Fq6* h[2] = {&e.c0, &e.c1};
    for (int k = 0; k < 2; ++k) {
      store_fq_canonical(h[k]->c0.c0, g + (k * 6 + 0) * 8);
      store_fq_canonical(h[k]->c0.c1, g + (k * 6 + 1) * 8);
      store_fq_canonical(h[k]->c1.c0, g + (k * 6 + 2) * 8);
      store_fq_canonical(h[k]->c1.c1, g + (k * 6 + 3) * 8);
      store_fq_canonical(h[k]->c2.c0, g + (k * 6 + 4) * 8);
      store_fq_canonical(h[k]->c2.c1, g + (k * 6 + 5) * 8);
    }
  }
}

int launch_g2_prepare(snarkv_ctx* ctx, const void* d_g2x2_256, void* d_prep) {
  hipLaunchKernelGGL(k_g2_prepare, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)d_g2x2_256,
                     (G2Prepared*)d_prep);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

int launch_validate_g2(snarkv_ctx* ctx, const void* d_g2x2_256, int* bad_host) {
  void* d_bad = nullptr;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_FLAGS, 64, &d_bad));
  SNARKV_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_validate_g2, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)d_g2x2_256, (int*)d_bad);
  SNARKV_HIP(hipGetLastError());
  SNARKV_HIP(hipMemcpyAsync(bad_host, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  return SNARKV_OK;
}

int launch_decide(snarkv_ctx* ctx, const void* d_prep, const void* d_accs, size_t m, void* d_ok, void* d_gt) {
  uint32_t blocks = (uint32_t)((m + 63) / 64);
  hipLaunchKernelGGL(k_decide, dim3(blocks), dim3(64), 0, ctx->stream, (const G2Prepared*)d_prep,
                     (const uint32_t*)d_accs, (uint32_t)m, (uint8_t*)d_ok, (uint32_t*)d_gt);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
