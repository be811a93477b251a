// Microbenchmark (dev tool): what an instruction of the k_accumulate mix COSTS on gfx950, in whole asm blocks
// (nothing the compiler can pad), at 1..4 resident waves per SIMD.  Answers, for the instruction diet of the
// accumulate loop: is `s_nop 0` free at 3 waves/SIMD, are VOP2 integer ops double-rate, what do a dependent
// v_mad_i64_i32 chain, an SGPR multiplier, a VGPR bank conflict and an SGPR-pair carry-out cost.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_issue tools/ubench_issue.hip && tools/ubench_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define R2(X) X X
#define R4(X) R2(X) R2(X)
#define R8(X) R4(X) R4(X)
#define R16(X) R8(X) R8(X)

// every probe: `iters` trips over a block of N instructions written out in one asm statement; fixed physical
// registers v0..v31 / s40..s51 (declared as clobbers), initialised inside the block's prologue statement
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","vcc"

#define PROBE(NAME, NINST, BODY)                                                                              \
  __global__ void __launch_bounds__(64) NAME(uint32_t* out, uint32_t a, int iters) {                          \
    uint32_t x = a + threadIdx.x;                                                                             \
    asm volatile("v_mov_b32 v0, %0\n v_mov_b32 v1, 0\n v_mov_b32 v2, %0\n v_mov_b32 v3, %1\n v_mov_b32 v4, %0\n"   \
                 "v_mov_b32 v5, 0\n v_mov_b32 v6, %1\n v_mov_b32 v7, %0\n v_mov_b32 v8, %1\n v_mov_b32 v9, 0\n"    \
                 "v_mov_b32 v10, %0\n v_mov_b32 v11, %1\n v_mov_b32 v12, %0\n v_mov_b32 v13, 0\n v_mov_b32 v14, %1\n" \
                 "v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, 0\n v_mov_b32 v18, %1\n v_mov_b32 v19, %0\n" \
                 "s_mov_b32 s44, 0x12345\n s_mov_b32 s45, 0x1abcdef\n s_mov_b32 s46, 0x1fffffff\n" ::"v"(x), "v"(a) : CLOB); \
    for (int i = 0; i < iters; ++i) asm volatile(BODY ::: CLOB);                                              \
    uint32_t r;                                                                                               \
    asm volatile("v_xor_b32 %0, v0, v1\n v_xor_b32 %0, %0, v4\n v_xor_b32 %0, %0, v5\n v_xor_b32 %0, %0, v8\n v_xor_b32 %0, %0, v12\n v_xor_b32 %0, %0, v16" : "=v"(r)::CLOB); \
    out[blockIdx.x * 64 + threadIdx.x] = r;                                                                   \
  }                                                                                                           \
  static const int NAME##_n = NINST;

#define MAD(acc, x, y) "v_mad_i64_i32 " acc ", vcc, " x ", " y ", " acc "\n"
#define MADS(acc, x, y) "v_mad_i64_i32 " acc ", s[40:41], " x ", " y ", " acc "\n"

// 1. dependent chain, operands in distinct banks (acc banks 0,1; src banks 2,3)
PROBE(mad_dep, 64, R16(R4(MAD("v[0:1]", "v2", "v3"))))
// 2. the same with a wave-level s_nop 0 after every mad
PROBE(mad_dep_nop_each, 64, R16(R4(MAD("v[0:1]", "v2", "v3") "s_nop 0\n")))
// 3. s_nop 0 after every 8th mad (the compiler's pattern is one per asm statement = per column)
PROBE(mad_dep_nop_8th, 64, R8(R4(MAD("v[0:1]", "v2", "v3")) R4(MAD("v[0:1]", "v2", "v3")) "s_nop 0\n"))
// 4. two independent chains interleaved
PROBE(mad_2chains, 64, R16(R2(MAD("v[0:1]", "v2", "v3") MAD("v[4:5]", "v6", "v7"))))
// 5. four independent chains
PROBE(mad_4chains, 64, R16(MAD("v[0:1]", "v2", "v3") MAD("v[4:5]", "v6", "v7") MAD("v[8:9]", "v10", "v11") MAD("v[12:13]", "v14", "v15")))
// 6. SGPR multiplier (the reduction products: m_j * p_(k-j) with p in SGPRs)
PROBE(mad_dep_sgpr, 64, R16(R4(MAD("v[0:1]", "v2", "s44"))))
// 7. bank conflict: both sources in bank 0 (with the accumulator's low word)
PROBE(mad_dep_bankconf, 64, R16(R4(MAD("v[0:1]", "v4", "v8"))))
// 8. carry-out to an SGPR pair instead of vcc
PROBE(mad_dep_scarry, 64, R16(R4(MADS("v[0:1]", "v2", "v3"))))
// 9. VOP2 / VOP3 ALU instructions of the mix, dependent chains
PROBE(and_dep, 64, R16(R4("v_and_b32 v0, 0x1fffffff, v0\n")))
PROBE(and_4chains, 64, R16("v_and_b32 v0, s46, v0\n v_and_b32 v4, s46, v4\n v_and_b32 v8, s46, v8\n v_and_b32 v12, s46, v12\n"))
PROBE(sub_4chains, 64, R16("v_sub_u32 v0, v0, v2\n v_sub_u32 v4, v4, v6\n v_sub_u32 v8, v8, v10\n v_sub_u32 v12, v12, v14\n"))
PROBE(ashr64_dep, 64, R16(R4("v_ashrrev_i64 v[0:1], 29, v[0:1]\n")))
PROBE(ashr64_4chains, 64, R16("v_ashrrev_i64 v[0:1], 29, v[0:1]\n v_ashrrev_i64 v[4:5], 29, v[4:5]\n v_ashrrev_i64 v[8:9], 29, v[8:9]\n v_ashrrev_i64 v[12:13], 29, v[12:13]\n"))
PROBE(mullo_dep, 64, R16(R4("v_mul_lo_u32 v0, v0, s44\n")))
PROBE(mullo_4chains, 64, R16("v_mul_lo_u32 v0, v0, s44\n v_mul_lo_u32 v4, v4, s44\n v_mul_lo_u32 v8, v8, s44\n v_mul_lo_u32 v12, v12, s44\n"))
PROBE(nop_only, 64, R16(R4("s_nop 0\n")))
// 9b. cross-lane moves of the latency-bound kernels (decide rounds, Poseidon): DPP adds, dependent on themselves
// (the required wait state before a DPP read of a fresh VGPR is written out as the compiler would) and over 4 / 9 registers
PROBE(dpp_quad_dep, 64, R16(R2("s_nop 1\n v_add_u32_dpp v0, v0, v0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")))
PROBE(dpp_quad_4regs, 64, R16("v_add_u32_dpp v0, v0, v0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v4, v4, v4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v8, v8, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v12, v12, v12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"))
PROBE(dpp_mirror_4regs, 64, R16("v_add_u32_dpp v0, v0, v0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v4, v4, v4 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v8, v8, v8 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v12, v12, v12 row_half_mirror row_mask:0xf bank_mask:0xf\n"))
PROBE(dpp_shr_4regs, 64, R16("v_mov_b32_dpp v0, v2 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp v4, v6 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp v8, v10 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp v12, v14 row_shr:4 row_mask:0xf bank_mask:0xa\n"))
// a mad result read by a DPP add two instructions later (product -> butterfly, as the kernels do)
PROBE(mad_then_dpp, 64, R16(MAD("v[0:1]", "v2", "v3") "v_and_b32 v4, s46, v4\n s_nop 0\n v_add_u32_dpp v8, v0, v8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"))
PROBE(mov_4chains, 64, R16("v_mov_b32 v0, v2\n v_mov_b32 v4, v6\n v_mov_b32 v8, v10\n v_mov_b32 v12, v14\n"))
PROBE(lshladd64_dep, 64, R16(R4("v_lshl_add_u64 v[0:1], v[0:1], 0, v[2:3]\n")))
// 10. one Montgomery COLUMN as the compiler emits it today (9 operand/reduction mads, nop, mul_lo, and, mad, nop, shift) ...
#define COL_TODAY                                                                                            \
  MAD("v[0:1]", "v2", "v3") MAD("v[0:1]", "v6", "v7") MAD("v[0:1]", "v10", "v11") MAD("v[0:1]", "v14", "v15") \
  MAD("v[0:1]", "v18", "v19") MAD("v[0:1]", "v16", "s44") MAD("v[0:1]", "v12", "s45") MAD("v[0:1]", "v8", "s44") MAD("v[0:1]", "v4", "s45") \
  "s_nop 0\n v_mul_lo_u32 v20, v0, s45\n v_and_b32 v20, 0x1fffffff, v20\n" MAD("v[0:1]", "v20", "s44") "s_nop 0\n v_ashrrev_i64 v[0:1], 29, v[0:1]\n"
PROBE(column_today, 15 * 4, R4(COL_TODAY))
// ... and without the two nops
#define COL_LEAN                                                                                             \
  MAD("v[0:1]", "v2", "v3") MAD("v[0:1]", "v6", "v7") MAD("v[0:1]", "v10", "v11") MAD("v[0:1]", "v14", "v15") \
  MAD("v[0:1]", "v18", "v19") MAD("v[0:1]", "v16", "s44") MAD("v[0:1]", "v12", "s45") MAD("v[0:1]", "v8", "s44") MAD("v[0:1]", "v4", "s45") \
  "v_mul_lo_u32 v20, v0, s45\n v_and_b32 v20, 0x1fffffff, v20\n" MAD("v[0:1]", "v20", "s44") "v_ashrrev_i64 v[0:1], 29, v[0:1]\n"
PROBE(column_lean, 13 * 4, R4(COL_LEAN))
// ... two columns of two independent products interleaved (ILP 2)
#define COL_LEAN_B                                                                                           \
  MAD("v[22:23]", "v2", "v3") MAD("v[22:23]", "v6", "v7") MAD("v[22:23]", "v10", "v11") MAD("v[22:23]", "v14", "v15") \
  MAD("v[22:23]", "v18", "v19") MAD("v[22:23]", "v16", "s44") MAD("v[22:23]", "v12", "s45") MAD("v[22:23]", "v8", "s44") MAD("v[22:23]", "v4", "s45") \
  "v_mul_lo_u32 v24, v22, s45\n v_and_b32 v24, 0x1fffffff, v24\n" MAD("v[22:23]", "v24", "s44") "v_ashrrev_i64 v[22:23], 29, v[22:23]\n"
PROBE(column_lean_x2, 26 * 2, R2(COL_LEAN COL_LEAN_B))

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
  printf("device %s, %d CUs, clock %.2f GHz; cycles = SIMD cycles per wave-instruction (counting every instruction of the block, s_nop included)\n", prop.name, cus, ghz);
  void* d;
  CHECK(hipMalloc(&d, 64 << 20));
  const int iters = 20000;
  printf("%-20s %8s %8s %8s %8s\n", "probe", "w=1", "w=2", "w=3", "w=4");
#define RUN(K)                                                                                              \
  {                                                                                                         \
    printf("%-20s", #K);                                                                                    \
    for (int w = 1; w <= 4; ++w) {                                                                          \
      int blocks = cus * 4 * w;                                                                             \
      double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(64), 0, 0, (uint32_t*)d, 12345u, iters); }); \
      double cyc = ms * 1e-3 * ghz * 1e9 / ((double)iters * K##_n * w);                                     \
      printf(" %8.2f", cyc);                                                                                \
    }                                                                                                       \
    printf("   (%d instr/block)\n", K##_n);                                                                 \
  }
  RUN(mad_dep) RUN(mad_dep_nop_each) RUN(mad_dep_nop_8th) RUN(mad_2chains) RUN(mad_4chains) RUN(mad_dep_sgpr)
  RUN(mad_dep_bankconf) RUN(mad_dep_scarry) RUN(and_dep) RUN(and_4chains) RUN(sub_4chains) RUN(ashr64_dep)
  RUN(ashr64_4chains) RUN(mullo_dep) RUN(mullo_4chains) RUN(nop_only) RUN(dpp_quad_dep) RUN(dpp_quad_4regs) RUN(dpp_mirror_4regs) RUN(dpp_shr_4regs) RUN(mad_then_dpp) RUN(mov_4chains) RUN(lshladd64_dep)
  RUN(column_today) RUN(column_lean) RUN(column_lean_x2)
  return 0;
}
