// Synthetic benchmark inputs generated IN HBM: seeded
// SplitMix64 streams -> uniform-ish canonical Fr scalars and G1 points.  A
// bench/test utility, not a reference function: the reference has no input
// generator on this path (its tests draw from OsRng / ChaCha20,
// snark-verifier/src/system/halo2/test.rs:191).  `oracle/c/bn254_oracle.c`
// restates the same streams so tests can check device inputs byte for byte.
#include "ctx.hpp"
#include "g1.h"
#include "g1_29.h"

namespace snarkv {

__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// 4 stream words, top two bits cleared (< 2^254), one conditional subtraction of r.
__global__ void k_sample_scalars(uint64_t seed, uint64_t first, uint32_t n, uint32_t* __restrict__ out, uint32_t mont) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr uint32_t r[8] = BN254_R_LIMBS;
  uint64_t s = seed ^ (0xA5A5A5A500000000ull + (first + i) * 0x9E3779B97F4A7C15ull);
  uint32_t w[8], d[8];
  for (int j = 0; j < 4; ++j) {
    uint64_t v = splitmix64(s);
    w[2 * j] = (uint32_t)v;
    w[2 * j + 1] = (uint32_t)(v >> 32);
  }
  w[7] &= 0x3FFFFFFFu;
  uint64_t borrow = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint64_t x = (uint64_t)w[j] - r[j] - borrow;
    d[j] = (uint32_t)x;
    borrow = (x >> 32) & 1u;
  }
  if (!borrow) {
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = d[j];
  }
  if (mont) {  // the same scalar in halo2curves' in-memory form: s * 2^256 mod r, by 256 modular doublings (set-up code)
    for (int t = 0; t < 256; ++t) {
      uint32_t c = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        uint32_t nw = (w[j] << 1) | c;
        c = w[j] >> 31;
        w[j] = nw;
      }
      borrow = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        uint64_t x = (uint64_t)w[j] - r[j] - borrow;
        d[j] = (uint32_t)x;
        borrow = (x >> 32) & 1u;
      }
      if (c || !borrow) {
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = d[j];
      }
    }
  }
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// x from the stream (< 2^252), incremented until x^3+3 is a square; p = 3 mod 4
// so y = (x^3+3)^((p+1)/4); the root with even canonical y is kept.
__global__ void __launch_bounds__(64) k_sample_points(uint64_t seed, uint64_t first, uint32_t n, uint32_t* __restrict__ out, uint32_t mont) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr uint32_t e[8] = BN254_P_PLUS_1_DIV_4_LIMBS;
  constexpr uint32_t three[8] = BN254_THREE_MONT;
  uint32_t ew[8];
  Fq b;
  for (int j = 0; j < 8; ++j) {
    ew[j] = e[j];
    b.v[j] = three[j];
  }
  uint64_t s = seed ^ (0x5A5A5A5A00000000ull + (first + i) * 0x9E3779B97F4A7C15ull);
  uint32_t w[8];
  for (int j = 0; j < 4; ++j) {
    uint64_t v = splitmix64(s);
    w[2 * j] = (uint32_t)v;
    w[2 * j + 1] = (uint32_t)(v >> 32);
  }
  w[7] &= 0x0FFFFFFFu;
  Fq x, y;
  for (;;) {
    x = fq_from_canonical(w);
    Fq rhs = fq_add(fq_mul(fq_sqr(x), x), b);
    y = fq_pow(rhs, ew);
    if (fq_eq(fq_sqr(y), rhs)) break;
    uint64_t c = (uint64_t)w[0] + 1;  // 64-bit increment, as the oracle
    w[0] = (uint32_t)c;
    w[1] += (uint32_t)(c >> 32);
  }
  uint32_t xw[8], yw[8];
  fq_to_canonical(x, xw);
  fq_to_canonical(y, yw);
  if (yw[0] & 1u) {
    y = fq_neg(y);
    fq_to_canonical(y, yw);
  }
  if (mont) {  // fq.h's domain is R = 2^256: its words ARE halo2curves' in-memory form
#pragma unroll
    for (int j = 0; j < 8; ++j) xw[j] = x.v[j], yw[j] = y.v[j];
  }
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 16);
  o[0] = make_uint4(xw[0], xw[1], xw[2], xw[3]);
  o[1] = make_uint4(xw[4], xw[5], xw[6], xw[7]);
  o[2] = make_uint4(yw[0], yw[1], yw[2], yw[3]);
  o[3] = make_uint4(yw[4], yw[5], yw[6], yw[7]);
}

// Integer-VALU roofline probe: every lane runs a dependent chain of Montgomery
// products (the hot loop's multiplier, fq29_mul) with 4 waves per SIMD on every
// CU -- the peak Fq-product rate the bucket-accumulate kernel can be held against.
__global__ void __launch_bounds__(256) k_ubench_fq29_mul(int32_t* __restrict__ out, uint32_t seed, int iters) {
  Fq29 x, y;
  for (int i = 0; i < 9; ++i) {
    x.v[i] = (int32_t)((seed * (threadIdx.x + 1) + i * 7919u) & 0x1FFFFFFFu);
    y.v[i] = (int32_t)(((seed ^ blockIdx.x) + i * 104729u) & 0x1FFFFFFFu);
  }
  x.v[8] &= 0xFFFFF;
  y.v[8] &= 0xFFFFF;
  for (int i = 0; i < iters; ++i) {
    x = fq29_mul(x, y);
    y = fq29_mul(y, x);
  }
  int32_t acc = 0;
  for (int i = 0; i < 9; ++i) acc ^= x.v[i] ^ y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// same for the whole mixed addition (8M + 2S + lazy adds + carry passes)
__global__ void __launch_bounds__(64, 4) k_ubench_madd29(int32_t* __restrict__ out, uint32_t seed, int iters) {
  G1Affine29 p;
  G1Xyzz29 acc;
  for (int i = 0; i < 9; ++i) {
    p.x.v[i] = (int32_t)((seed * (threadIdx.x + 1) + i * 7919u) & 0x1FFFFFFFu);
    p.y.v[i] = (int32_t)(((seed ^ blockIdx.x) + i * 104729u) & 0x1FFFFFFFu);
    acc.x.v[i] = (int32_t)((i * 31u + threadIdx.x) & 0x1FFFFFFFu);
    acc.y.v[i] = 3 * i + 1;
    acc.zz.v[i] = 5 * i + 2;
    acc.zzz.v[i] = 7 * i + 3;
  }
  p.x.v[8] &= 0xFFFFF;
  p.y.v[8] &= 0xFFFFF;
  for (int i = 0; i < iters; ++i) {
    xyzz29_madd_fast(acc, p);
    p.x.v[0] = (p.x.v[0] + 1) & 0x1FFFFFFF;
  }
  int32_t a = 0;
  for (int i = 0; i < 9; ++i) a ^= acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int launch_ubench(snarkv_ctx* ctx, int which, int iters, double* ops_per_s) {
  hipDeviceProp_t prop;
  SNARKV_HIP(hipGetDeviceProperties(&prop, ctx->device));
  int cus = prop.multiProcessorCount;
  void* d_out;
  uint32_t blocks = which == 0 ? (uint32_t)cus * 4u : (uint32_t)cus * 16u;  // 4 waves per SIMD either way
  uint32_t threads = which == 0 ? 256u : 64u;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_MISC2, (size_t)blocks * threads * 4, &d_out));
  hipEvent_t a, b;
  SNARKV_HIP(hipEventCreate(&a));
  SNARKV_HIP(hipEventCreate(&b));
  double best = 0;
  for (int rep = 0; rep < 3; ++rep) {
    SNARKV_HIP(hipEventRecord(a, ctx->stream));
    if (which == 0)
      hipLaunchKernelGGL(k_ubench_fq29_mul, dim3(blocks), dim3(threads), 0, ctx->stream, (int32_t*)d_out, 77u + rep, iters);
    else
      hipLaunchKernelGGL(k_ubench_madd29, dim3(blocks), dim3(threads), 0, ctx->stream, (int32_t*)d_out, 77u + rep, iters);
    SNARKV_HIP(hipEventRecord(b, ctx->stream));
    SNARKV_HIP(hipEventSynchronize(b));
    float ms = 0;
    SNARKV_HIP(hipEventElapsedTime(&ms, a, b));
    double ops = (double)blocks * threads * (double)iters * (which == 0 ? 2.0 : 1.0) / (ms * 1e-3);
    if (ops > best) best = ops;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *ops_per_s = best;
  return SNARKV_OK;
}

int launch_sample_scalars(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_out) {
  hipLaunchKernelGGL(k_sample_scalars, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, first,
                     (uint32_t)n, (uint32_t*)d_out, ctx->mont ? 1u : 0u);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

int launch_sample_points(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_out) {
  hipLaunchKernelGGL(k_sample_points, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, ctx->stream, seed, first,
                     (uint32_t)n, (uint32_t*)d_out, ctx->mont ? 1u : 0u);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
