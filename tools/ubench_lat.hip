// Microbenchmark (dev tool): SINGLE-WAVEFRONT latency of the field products the
// latency-bound kernels (k_decide rounds, small-MSM doubling chains) are made
// of: one wavefront per CU, dependent chain of `iters` operations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_lat tools/ubench_lat.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../snark-verifier_amd/csrc/fq29.h"

using namespace snarkv;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// column-parallel form: 17 independent column chains, then operand-scanning reduction
template <bool TWO>
__device__ __forceinline__ Fq29 mul_cols(const Fq29& a, const Fq29& b, const Fq29& c, const Fq29& d) {
  int64_t col[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) col[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      col[i + j] += (int64_t)a.v[i] * b.v[j];
      if (TWO) col[i + j] += (int64_t)c.v[i] * d.v[j];
    }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    int32_t m = (int32_t)(((uint32_t)col[k] * (uint32_t)BN254_P29_NINV) & (uint32_t)kMask29);
#pragma unroll
    for (int j = 0; j < 9; ++j) col[k + j] += (int64_t)m * fq29_p(j);
    col[k + 1] += col[k] >> 29;
  }
  Fq29 r;
  int64_t acc = col[9];
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    r.v[k - 9] = (int32_t)acc & kMask29;
    acc = (acc >> 29) + (k + 1 < 17 ? col[k + 1] : 0);
  }
  r.v[8] = (int32_t)acc;
  return r;
}

template <int WHICH>
__global__ void __launch_bounds__(64) k_lat(int32_t* out, uint32_t seed, int iters) {
  Fq29 x, y, z;
  for (int i = 0; i < 9; ++i) {
    x.v[i] = (int32_t)((seed * (threadIdx.x + 1) + i * 7919u) & 0x1FFFFFFFu);
    y.v[i] = (int32_t)(((seed ^ blockIdx.x) + i * 104729u) & 0x1FFFFFFFu);
    z.v[i] = (int32_t)(((seed + 5) * (threadIdx.x + 3) + i * 31u) & 0x1FFFFFFFu);
  }
  x.v[8] &= 0xFFFFF; y.v[8] &= 0xFFFFF; z.v[8] &= 0xFFFFF;
  for (int i = 0; i < iters; ++i) {
    if (WHICH == 0) x = fq29_mul(x, y);
    if (WHICH == 1) x = mul_cols<false>(x, y, x, y);
    if (WHICH == 2) x = fq29_mul2(x, y, z, x);
    if (WHICH == 3) x = mul_cols<true>(x, y, z, x);
    if (WHICH == 4) x = fq29_sqr(x);
  }
  int32_t acc = 0;
  for (int i = 0; i < 9; ++i) acc ^= x.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int CH>
__global__ void __launch_bounds__(64) k_madchain(uint64_t* out, uint32_t a, uint32_t b, int iters) {
  uint64_t r[8];
  for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;
  uint32_t va = a + threadIdx.x, vb = b ^ threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32 / CH; ++k)
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[c]) : "v"(va), "v"(vb) : "vcc");
  }
  uint64_t s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  void* d;
  CHECK(hipMalloc(&d, 1 << 22));
  const int iters = 4000;
  const char* names[5] = {"fq29_mul (product scanning)", "mul_cols<1 product>", "fq29_mul2 (product scanning)", "mul_cols<2 products>", "fq29_sqr"};
  double ms;
  ms = time_ms([&] { hipLaunchKernelGGL(k_lat<0>, dim3(256), dim3(64), 0, 0, (int32_t*)d, 7u, iters); });
  printf("%-32s %8.1f ns/op (1 wave/CU)\n", names[0], ms * 1e6 / iters);
  ms = time_ms([&] { hipLaunchKernelGGL(k_lat<1>, dim3(256), dim3(64), 0, 0, (int32_t*)d, 7u, iters); });
  printf("%-32s %8.1f ns/op (1 wave/CU)\n", names[1], ms * 1e6 / iters);
  ms = time_ms([&] { hipLaunchKernelGGL(k_lat<2>, dim3(256), dim3(64), 0, 0, (int32_t*)d, 7u, iters); });
  printf("%-32s %8.1f ns/op (1 wave/CU)\n", names[2], ms * 1e6 / iters);
  ms = time_ms([&] { hipLaunchKernelGGL(k_lat<3>, dim3(256), dim3(64), 0, 0, (int32_t*)d, 7u, iters); });
  printf("%-32s %8.1f ns/op (1 wave/CU)\n", names[3], ms * 1e6 / iters);
  ms = time_ms([&] { hipLaunchKernelGGL(k_lat<4>, dim3(256), dim3(64), 0, 0, (int32_t*)d, 7u, iters); });
  printf("%-32s %8.1f ns/op (1 wave/CU)\n", names[4], ms * 1e6 / iters);
  const int it2 = 20000;
  ms = time_ms([&] { hipLaunchKernelGGL(k_madchain<1>, dim3(256), dim3(64), 0, 0, (uint64_t*)d, 3u, 5u, it2); });
  printf("v_mad_u64_u32 1 chain  : %6.2f ns/instr\n", ms * 1e6 / (it2 * 32.0));
  ms = time_ms([&] { hipLaunchKernelGGL(k_madchain<2>, dim3(256), dim3(64), 0, 0, (uint64_t*)d, 3u, 5u, it2); });
  printf("v_mad_u64_u32 2 chains : %6.2f ns/instr\n", ms * 1e6 / (it2 * 32.0));
  ms = time_ms([&] { hipLaunchKernelGGL(k_madchain<4>, dim3(256), dim3(64), 0, 0, (uint64_t*)d, 3u, 5u, it2); });
  printf("v_mad_u64_u32 4 chains : %6.2f ns/instr\n", ms * 1e6 / (it2 * 32.0));
  ms = time_ms([&] { hipLaunchKernelGGL(k_madchain<8>, dim3(256), dim3(64), 0, 0, (uint64_t*)d, 3u, 5u, it2); });
  printf("v_mad_u64_u32 8 chains : %6.2f ns/instr\n", ms * 1e6 / (it2 * 32.0));
  return 0;
}
