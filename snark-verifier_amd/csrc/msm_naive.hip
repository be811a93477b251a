// Batched small-MSM kernels: the device side of
// `NativeLoader::multi_scalar_multiplication`
// (reference snark-verifier/src/loader/native.rs:61-71).
//
// The accumulation path produces MANY independent small MSMs (per proof a
// ~21-term and a ~3-term one from Gwc19/Bdfg21::verify, then two (m+1)-term
// ones from KzgAs::verify -- SURVEY.md section 0 item 6), so the launch unit
// is a SEGMENTED MSM:
//   K1  k_term_scalar_mul : one lane per (scalar, base) term, 256-step
//                           double-and-add in XYZZ (`*base * scalar`, native.rs:67)
//   K2  k_segment_fold    : one wave per MSM folds its terms (`reduce(|a,v| a+v)`,
//                           native.rs:68), then `to_affine()` (native.rs:70) and
//                           canonical little-endian store.
#include "ctx.hpp"
#include "g1.cuh"

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

__global__ void __launch_bounds__(64) k_term_scalar_mul(const uint32_t* __restrict__ scalars,
                                                         const uint32_t* __restrict__ points,
                                                         G1Xyzz* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8], pw[16];
  load_words16(scalars + (size_t)i * 8, k, 2);
  load_words16(points + (size_t)i * 16, pw, 4);
  G1Affine p = g1a_from_canonical(pw);
  out[i] = g1_scalar_mul(p, k);
}

// One 64-lane block (= one wavefront) per MSM.
__global__ void __launch_bounds__(64) k_segment_fold(const G1Xyzz* __restrict__ terms,
                                                      const uint32_t* __restrict__ offsets,
                                                      uint32_t* __restrict__ out) {
  __shared__ G1Xyzz sh[64];
  uint32_t k = blockIdx.x;
  uint32_t lo = offsets[k], hi = offsets[k + 1];
  uint32_t lane = threadIdx.x;
  G1Xyzz acc = xyzz_identity();
  for (uint32_t i = lo + lane; i < hi; i += 64) xyzz_add(acc, terms[i]);
  sh[lane] = acc;
  __syncthreads();
  for (uint32_t s = 32; s >= 1; s >>= 1) {
    if (lane < s) {
      G1Xyzz a = sh[lane];
      xyzz_add(a, sh[lane + s]);
      sh[lane] = a;
    }
    __syncthreads();
  }
  if (lane == 0) {
    G1Affine r = xyzz_to_affine(sh[0]);
    uint32_t w[16];
    g1a_to_canonical(r, w);
    uint4* o = reinterpret_cast<uint4*>(out + (size_t)k * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
  }
}

// Optional input validation (SNARKV_FLAG_VALIDATE): canonical scalars (< r),
// canonical coordinates (< p), on-curve.  bad[0] counts offenders.
__global__ void k_validate(const uint32_t* __restrict__ scalars, const uint32_t* __restrict__ points, uint32_t n,
                           int* __restrict__ bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = true;
  if (scalars) {
    constexpr uint32_t r[8] = BN254_R_LIMBS;
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
    if (ok) ok = g1a_is_on_curve(g1a_from_canonical(pw));
  }
  if (!ok) atomicAdd(bad, 1);
}

int launch_msm_batched(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, const void* d_offsets,
                       size_t n_msm, size_t n_terms, void* d_out) {
  void* d_terms = nullptr;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_TERM_PARTIALS, n_terms * sizeof(G1Xyzz), &d_terms));
  uint32_t blocks = (uint32_t)((n_terms + 63) / 64);
  hipLaunchKernelGGL(k_term_scalar_mul, dim3(blocks), dim3(64), 0, ctx->stream, (const uint32_t*)d_scalars,
                     (const uint32_t*)d_points, (G1Xyzz*)d_terms, (uint32_t)n_terms);
  hipLaunchKernelGGL(k_segment_fold, dim3((uint32_t)n_msm), dim3(64), 0, ctx->stream, (const G1Xyzz*)d_terms,
                     (const uint32_t*)d_offsets, (uint32_t*)d_out);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

int launch_validate(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int* bad_host) {
  void* d_bad = nullptr;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_FLAGS, 64, &d_bad));
  SNARKV_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
  uint32_t blocks = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(k_validate, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t*)d_scalars,
                     (const uint32_t*)d_points, (uint32_t)n, (int*)d_bad);
  SNARKV_HIP(hipGetLastError());
  SNARKV_HIP(hipMemcpyAsync(bad_host, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  return SNARKV_OK;
}

}  // namespace snarkv
