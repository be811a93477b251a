// Microbenchmark (dev tool): issue rate of the integer / fp64 instructions a
// 254-bit Montgomery multiplier can be built from on gfx950, and the
// throughput of the shipped fq_mul / G1 adders.  Prints one line per probe.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_fq tools/ubench_fq.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../snark-verifier_amd/csrc/g1.h"

using namespace snarkv;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// 8 independent chains x 4 = 32 instructions per loop body
#define PROBE_KERNEL(NAME, ASM_LINE, TYPE, CONSTR_ACC)                                   \
  __global__ void NAME(TYPE* out, uint32_t a, uint32_t b, int iters) {                   \
    TYPE r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    uint32_t va = a + threadIdx.x, vb = b ^ threadIdx.x;                                 \
    for (int i = 0; i < iters; ++i) {                                                    \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                    \
        asm volatile(ASM_LINE : CONSTR_ACC(r0) : "v"(va), "v"(vb) : "vcc");              \
        asm volatile(ASM_LINE : CONSTR_ACC(r1) : "v"(va), "v"(vb) : "vcc");              \
        asm volatile(ASM_LINE : CONSTR_ACC(r2) : "v"(va), "v"(vb) : "vcc");              \
        asm volatile(ASM_LINE : CONSTR_ACC(r3) : "v"(va), "v"(vb) : "vcc");              \
        asm volatile(ASM_LINE : CONSTR_ACC(r4) : "v"(va), "v"(vb) : "vcc");              \
        asm volatile(ASM_LINE : CONSTR_ACC(r5) : "v"(va), "v"(vb) : "vcc");              \
        asm volatile(ASM_LINE : CONSTR_ACC(r6) : "v"(va), "v"(vb) : "vcc");              \
        asm volatile(ASM_LINE : CONSTR_ACC(r7) : "v"(va), "v"(vb) : "vcc");              \
      }                                                                                  \
    }                                                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;  \
  }

#define ACC_RW(x) "+v"(x)

PROBE_KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0", uint64_t, ACC_RW)
PROBE_KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0", uint32_t, ACC_RW)
PROBE_KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %0, %1, %0", uint32_t, ACC_RW)
PROBE_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0", uint32_t, ACC_RW)
PROBE_KERNEL(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %1, %0", uint32_t, ACC_RW)
PROBE_KERNEL(k_add_u32, "v_add_u32 %0, %1, %0", uint32_t, ACC_RW)
PROBE_KERNEL(k_add_co_u32, "v_add_co_u32 %0, vcc, %1, %0", uint32_t, ACC_RW)
PROBE_KERNEL(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %1, %0, vcc", uint32_t, ACC_RW)
PROBE_KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %0", uint64_t, ACC_RW)
PROBE_KERNEL(k_mov_b32, "v_mov_b32 %0, %1", uint32_t, ACC_RW)
PROBE_KERNEL(k_add3_u32, "v_add3_u32 %0, %1, %2, %0", uint32_t, ACC_RW)
PROBE_KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %0, 3", uint32_t, ACC_RW)
PROBE_KERNEL(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0", uint64_t, ACC_RW)

__global__ void k_fma_f64(double* out, uint32_t a, uint32_t b, int iters) {
  double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
  double va = 1.0 + a * 1e-9, vb = b * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r0) : "v"(va), "v"(vb));
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r1) : "v"(va), "v"(vb));
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r2) : "v"(va), "v"(vb));
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r3) : "v"(va), "v"(vb));
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r4) : "v"(va), "v"(vb));
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r5) : "v"(va), "v"(vb));
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r6) : "v"(va), "v"(vb));
      asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r7) : "v"(va), "v"(vb));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

// dependent chain of Montgomery products (the shipped fq_mul)
__global__ void __launch_bounds__(256) k_fq_mul_chain(uint32_t* out, uint32_t seed, int iters) {
  Fq x, y;
  for (int i = 0; i < 8; ++i) {
    x.v[i] = seed * (threadIdx.x + 1) + i;
    y.v[i] = seed ^ (blockIdx.x + i);
  }
  x.v[7] &= 0x0FFFFFFF;
  y.v[7] &= 0x0FFFFFFF;
  for (int i = 0; i < iters; ++i) {
    x = fq_mul(x, y);
    y = fq_mul(y, x);
  }
  uint32_t acc = 0;
  for (int i = 0; i < 8; ++i) acc ^= x.v[i] ^ y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_fq_addsub_chain(uint32_t* out, uint32_t seed, int iters) {
  Fq x, y;
  for (int i = 0; i < 8; ++i) {
    x.v[i] = seed * (threadIdx.x + 1) + i;
    y.v[i] = seed ^ (blockIdx.x + i);
  }
  x.v[7] &= 0x0FFFFFFF;
  y.v[7] &= 0x0FFFFFFF;
  for (int i = 0; i < iters; ++i) {
    x = fq_add(x, y);
    y = fq_sub(y, x);
  }
  uint32_t acc = 0;
  for (int i = 0; i < 8; ++i) acc ^= x.v[i] ^ y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// chain of mixed additions (the bucket-accumulate inner step)
__global__ void __launch_bounds__(64) k_madd_chain(uint32_t* out, uint32_t seed, int iters) {
  G1Affine p;
  G1Xyzz acc;
  for (int i = 0; i < 8; ++i) {
    p.x.v[i] = seed * (threadIdx.x + 1) + i;
    p.y.v[i] = seed ^ (blockIdx.x + i);
    acc.x.v[i] = i + threadIdx.x;
    acc.y.v[i] = 3 * i + 1;
    acc.zz.v[i] = 5 * i + 2;
    acc.zzz.v[i] = 7 * i + 3;
  }
  p.x.v[7] &= 0x0FFFFFFF; p.y.v[7] &= 0x0FFFFFFF;
  acc.x.v[7] = 1; acc.y.v[7] = 2; acc.zz.v[7] = 3; acc.zzz.v[7] = 4;
  for (int i = 0; i < iters; ++i) {
    xyzz_add_mixed(acc, p);
    p.x.v[0] += 1;
  }
  uint32_t a = 0;
  for (int i = 0; i < 8; ++i) a ^= acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <typename F>
static double time_ms(F launch, int reps = 3) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
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
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double ghz = prop.clockRate / 1e6;
  printf("device %s, %d CUs, clock %.2f GHz\n", prop.name, cus, ghz);
  void* d;
  CHECK(hipMalloc(&d, 64 << 20));
  const int iters = 4000;
#define RUN_PROBE(K, TYPE, WAVES_PER_SIMD)                                                     \
  {                                                                                            \
    int blocks = cus * WAVES_PER_SIMD;                                                         \
    double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, (TYPE*)d, 12345u, 6789u, iters); }); \
    double winst = (double)blocks * 4 * iters * 32;                                            \
    double per_simd_cycle = winst / (cus * 4.0) / (ms * 1e-3 * ghz * 1e9);                      \
    printf("%-18s waves/SIMD=%d  %.3f ms  %.3f wave-inst/cycle/SIMD  => %.2f cycles/inst\n", #K, WAVES_PER_SIMD, ms, \
           per_simd_cycle, 1.0 / per_simd_cycle);                                              \
  }
  for (int w : {1, 2, 4}) {
    RUN_PROBE(k_mad_u64_u32, uint64_t, w);
    RUN_PROBE(k_mul_lo_u32, uint32_t, w);
    RUN_PROBE(k_mul_hi_u32, uint32_t, w);
    RUN_PROBE(k_mad_u32_u24, uint32_t, w);
    RUN_PROBE(k_mul_hi_u32_u24, uint32_t, w);
    RUN_PROBE(k_add_u32, uint32_t, w);
    RUN_PROBE(k_add_co_u32, uint32_t, w);
    RUN_PROBE(k_addc_co_u32, uint32_t, w);
    RUN_PROBE(k_lshl_add_u64, uint64_t, w);
    RUN_PROBE(k_mov_b32, uint32_t, w);
    RUN_PROBE(k_add3_u32, uint32_t, w);
    RUN_PROBE(k_alignbit, uint32_t, w);
    RUN_PROBE(k_lshrrev_b64, uint64_t, w);
    RUN_PROBE(k_fma_f64, double, w);
  }
  for (int w : {1, 2, 3, 4}) {
    int blocks = cus * w;
    int it = 2000;
    double ms = time_ms([&] { hipLaunchKernelGGL(k_fq_mul_chain, dim3(blocks), dim3(256), 0, 0, (uint32_t*)d, 77u, it); });
    double muls = (double)blocks * 256 * it * 2;
    printf("fq_mul chain      waves/SIMD=%d  %.3f ms  %.3e Fq-mul/s  (%.0f SIMD-cycles per wave-mul)\n", w, ms,
           muls / (ms * 1e-3), (ms * 1e-3 * ghz * 1e9) / ((double)it * 2 * w));
    ms = time_ms([&] { hipLaunchKernelGGL(k_fq_addsub_chain, dim3(blocks), dim3(256), 0, 0, (uint32_t*)d, 77u, it); });
    printf("fq_add/sub chain  waves/SIMD=%d  %.3f ms  (%.0f SIMD-cycles per wave-op)\n", w, ms,
           (ms * 1e-3 * ghz * 1e9) / ((double)it * 2 * w));
  }
  for (int w : {1, 2, 3}) {
    int blocks = cus * 4 * w;
    int it = 300;
    double ms = time_ms([&] { hipLaunchKernelGGL(k_madd_chain, dim3(blocks), dim3(64), 0, 0, (uint32_t*)d, 77u, it); });
    double adds = (double)blocks * 64 * it;
    printf("xyzz madd chain   waves/SIMD=%d  %.3f ms  %.3e madd/s  (%.0f SIMD-cycles per wave-madd)\n", w, ms,
           adds / (ms * 1e-3), (ms * 1e-3 * ghz * 1e9) / ((double)it * w));
  }
  return 0;
}
