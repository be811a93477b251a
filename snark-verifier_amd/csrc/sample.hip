// Synthetic benchmark inputs generated IN HBM: seeded
// SplitMix64 streams -> uniform-ish canonical Fr scalars and G1 points.  A
// bench/test utility, not a reference function: the reference has no input
// generator on this path (its tests draw from OsRng / ChaCha20,
// snark-verifier/src/system/halo2/test.rs:191).  `oracle/c/bn254_oracle.c`
// restates the same streams so tests can check device inputs byte for byte.
#include "ctx.hpp"
#include "g1.cuh"

namespace snarkv {

__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// 4 stream words, top two bits cleared (< 2^254), one conditional subtraction of r.
__global__ void k_sample_scalars(uint64_t seed, uint64_t first, uint32_t n, uint32_t* __restrict__ out) {
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
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 8);
  if (borrow) {
    o[0] = make_uint4(w[0], w[1], w[2], w[3]);
    o[1] = make_uint4(w[4], w[5], w[6], w[7]);
  } else {
    o[0] = make_uint4(d[0], d[1], d[2], d[3]);
    o[1] = make_uint4(d[4], d[5], d[6], d[7]);
  }
}

// x from the stream (< 2^252), incremented until x^3+3 is a square; p = 3 mod 4
// so y = (x^3+3)^((p+1)/4); the root with even canonical y is kept.
__global__ void __launch_bounds__(64) k_sample_points(uint64_t seed, uint64_t first, uint32_t n, uint32_t* __restrict__ out) {
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
  if (yw[0] & 1u) fq_to_canonical(fq_neg(y), yw);
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)i * 16);
  o[0] = make_uint4(xw[0], xw[1], xw[2], xw[3]);
  o[1] = make_uint4(xw[4], xw[5], xw[6], xw[7]);
  o[2] = make_uint4(yw[0], yw[1], yw[2], yw[3]);
  o[3] = make_uint4(yw[4], yw[5], yw[6], yw[7]);
}

int launch_sample_scalars(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_out) {
  hipLaunchKernelGGL(k_sample_scalars, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, first,
                     (uint32_t)n, (uint32_t*)d_out);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

int launch_sample_points(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_out) {
  hipLaunchKernelGGL(k_sample_points, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, ctx->stream, seed, first,
                     (uint32_t)n, (uint32_t*)d_out);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
