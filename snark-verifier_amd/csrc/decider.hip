// KZG pairing decider on the device.
// Replaces the native `AccumulationDecider` impl for `KzgAs`
// (reference snark-verifier/src/pcs/kzg/decider.rs:70-93):
//   decide     : e(lhs, g2) * e(rhs, -s_g2) == 1
//   decide_all : the same for every accumulator of a list (decider.rs:84-93)
//
//   D0 k_g2_prepare_w : once per deciding key -- line tables of g2 and -s_g2
//                     (the reference's `G2Prepared::from`, redone there on
//                     every call, decider.rs:74), one wavefront per key (g2_prepare_w.h)
//   D1 k_decide     : ONE 128-lane WORKGROUP per accumulator (pairing_coop29.h):
//                     2-pair Miller loop with shared squarings + exact final
//                     exponentiation + `is_identity`; every Fq12 product is one
//                     parallel round of 72 fused two-product Montgomery steps,
//                     summed by DPP butterflies inside 8-lane groups (no LDS
//                     traffic for the reduction, one store + two barriers).
//                     (v1 gave one lane per accumulator: 45-55 ms per decide,
//                     whatever the batch size; v2 = 204 products + two LDS
//                     reduction stages: 1.64 ms.)
// Batches of independent accumulators (decide_all) are the second parallel axis
// (SURVEY.md 8e).
#include "ctx.hpp"
#include "g1.h"
#include "pairing.h"
#include "pairing_coop.h"
#include "pairing_coop29.h"
#include "decide_w.h"
#include "decide_sched.hpp"
#include "g2_prepare_w.h"
#include <mutex>

namespace snarkv {

size_t g2_prepared_bytes() { return sizeof(G2Prepared29); }  // per point; a key holds two (g2, -s_g2)
static size_t prep29_offset() { return 0; }

// y^2 == x^3 + 3/(9+u) and canonical coordinates, for both G2 points (lanes 0, 1), on the lazy 29-bit field (all inline:
// the tower functions of tower.h take their operands by reference and would put them on the stack)
__global__ void __launch_bounds__(64) k_validate_g2(const uint32_t* __restrict__ g2x2, int* __restrict__ bad, uint32_t mont) {
  uint32_t k = threadIdx.x;
  if (k >= 2) return;
  const uint32_t* src = g2x2 + k * 32;
  bool ok = true;
  for (int j = 0; j < 4; ++j) ok = ok && fq_canonical_in_range(src + 8 * j);  // (the in-memory form is reduced as well)
  if (ok) {
    Fq29 x0 = fq29_canon_residue(fq29_from_words(src, mont != 0)), x1 = fq29_canon_residue(fq29_from_words(src + 8, mont != 0));
    Fq29 y0 = fq29_canon_residue(fq29_from_words(src + 16, mont != 0)), y1 = fq29_canon_residue(fq29_from_words(src + 24, mont != 0));
    const bool id = fq29_limbs_all_zero(x0) && fq29_limbs_all_zero(x1) && fq29_limbs_all_zero(y0) && fq29_limbs_all_zero(y1);
    if (!id) {
      constexpr uint32_t bc[2][8] = {BN254_TWIST_B_C0_MONT, BN254_TWIST_B_C1_MONT};
      auto mul = [](const Fq29& a0, const Fq29& a1, const Fq29& b0, const Fq29& b1, Fq29& r0, Fq29& r1) {
        r0 = fq29_mul2(a0, b0, fq29_neg(a1), b1);
        r1 = fq29_mul2(a0, b1, a1, b0);
      };
      Fq29 l0, l1, s0, s1, c0, c1;
      mul(y0, y1, y0, y1, l0, l1);                                  // y^2
      mul(x0, x1, x0, x1, s0, s1);                                  // x^2
      mul(fq29_norm(s0), fq29_norm(s1), x0, x1, c0, c1);            // x^3
      const Fq29 r0 = fq29_add(c0, g2w_from_mont32(bc[0])), r1 = fq29_add(c1, g2w_from_mont32(bc[1]));
      ok = fq29_is_zero_mod_p(fq29_sub(l0, r0)) && fq29_is_zero_mod_p(fq29_sub(l1, r1));
    }
  }
  if (!ok) atomicAdd(bad, 1);
}

// ------------------------------------------------------------------ D1
constexpr int kDecideThreads = 128;  // 96 lanes carry the round (pairing_coop29.h coop3), 2 wavefronts

struct CoopReg {  // an Fq12 in the flat basis (c = 2i+e <-> u^e w^i)
  Fq29 v[12];
};

enum { RF = 0, RT, RINV, RFX, RFX2, RFX3, RY0, RY1, RY2, RY3, RY4, RY5, RY6, RT0, RT1, RONE, RZERO, RCOUNT };

struct CoopShared {
  CoopReg r[RCOUNT];
  Fq29 pt[2][2];                  // (x, y) of lhs, rhs
  Fq29 scal[2];                   // an Fq2 scalar (inverse of the norm)
  int live[2];
};

// One workgroup = one accumulator; the whole state lives in LDS (file scope so
// that every helper addresses it with ds_* instructions, not flat pointers).
__shared__ CoopShared g_sh;
// line tables evaluated at this accumulator's points: l0 = cy*yP, l1 = cx*xP per pair and line
// (l2 = cw is read from the key: 29 KB, four workgroups per CU)
__shared__ Fq29 g_lines4[2][kLinesPerG2][4];

static __device__ __forceinline__ void coop_store(int dst, int c, const Fq29& val) { g_sh.r[dst].v[c] = val; }

// DPP lane exchange inside a 16-lane row (no LDS, no barrier)
template <int CTRL>
static __device__ __forceinline__ uint32_t dpp_u32(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}

// every lane of an aligned 8-lane group gets the limb-wise sum over the group
// (limbs as unsigned: <= 6 live terms of < 2^29 each)
static __device__ __forceinline__ Fq29 group8_sum(Fq29 x) {
#pragma unroll
  for (int i = 0; i < 9; ++i) x.v[i] = (int32_t)((uint32_t)x.v[i] + dpp_u32<0xB1>((uint32_t)x.v[i]));   // quad_perm [1,0,3,2]
#pragma unroll
  for (int i = 0; i < 9; ++i) x.v[i] = (int32_t)((uint32_t)x.v[i] + dpp_u32<0x4E>((uint32_t)x.v[i]));   // quad_perm [2,3,0,1]
#pragma unroll
  for (int i = 0; i < 9; ++i) x.v[i] = (int32_t)((uint32_t)x.v[i] + dpp_u32<0x141>((uint32_t)x.v[i]));  // row_half_mirror
  return x;
}

// dst = a * b, b a register (b_reg >= 0) or the sparse line (pair, idx): only
// w^0, w^1, w^3 non-zero.  All lanes of the workgroup call it.
static __device__ __forceinline__ void coop_mul_b(int dst, int a, int b_reg, int pair, int idx,
                                                  const G2Prepared29* __restrict__ prep) {
  const int tid = threadIdx.x;
  const Coop3Lane L = coop3_lane(tid);
  Fq29 val = fq29_zero();
  if (L.active) {
    const Fq29& a0 = g_sh.r[a].v[2 * L.i1];
    const Fq29& a1 = g_sh.r[a].v[2 * L.i1 + 1];
    const int c0 = 2 * L.i2 + L.e, c1 = 2 * L.i2 + 1 - L.e;
    if (b_reg >= 0) {
      val = coop3_product(L.e, a0, a1, g_sh.r[b_reg].v[c0], g_sh.r[b_reg].v[c1]);
    } else if (L.i2 < 2 || L.i2 == 3) {
      // one product stream for the three non-zero line coefficients: l0, l1 from LDS, l2 = cw from the key
      Fq29 y0, y1;
      if (L.i2 < 2) {
        y0 = g_lines4[pair][idx][c0];
        y1 = g_lines4[pair][idx][c1];
      } else {
        y0 = prep[pair].line[idx].c[4 + L.e];
        y1 = prep[pair].line[idx].c[5 - L.e];
      }
      val = coop3_product(L.e, a0, a1, y0, y1);
    }
  }
  Fq29 lo, hi;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    lo.v[i] = L.high ? 0 : val.v[i];
    hi.v[i] = L.high ? val.v[i] : 0;
  }
  lo = group8_sum(lo);
  hi = group8_sum(hi);
  Fq29 hp;  // the high sum of the other u-power: 8 lanes away in the same row
#pragma unroll
  for (int i = 0; i < 9; ++i) hp.v[i] = (int32_t)dpp_u32<0x128>((uint32_t)hi.v[i]);  // row_ror:8
  const bool writer = (tid & 7) == 0 && L.k < 6;
  Fq29 res = fq29_zero();
  if (writer) res = coop3_finalize(L.e, lo, hi, hp);
  __syncthreads();  // every operand read is done before dst (possibly == a or b) changes
  if (writer) coop_store(dst, 2 * L.k + L.e, res);
  __syncthreads();
}

static __device__ __noinline__ void coop_mulr(int dst, int a, int b) { coop_mul_b(dst, a, b, 0, 0, nullptr); }
static __device__ __noinline__ void coop_mull(int pair, int idx, const G2Prepared29* __restrict__ prep) {
  coop_mul_b(RF, RF, -1, pair, idx, prep);
}

// dst = conj(a): negate the odd powers of w (the c1 half of the tower)
static __device__ __noinline__ void coop_conj(int dst, int a) {
  int tid = threadIdx.x;
  if (tid < 12) {
    bool odd = ((tid >> 1) & 1) != 0;
    Fq29 x = g_sh.r[a].v[tid];
    g_sh.r[dst].v[tid] = odd ? fq29_norm(fq29_neg(x)) : x;
  }
  __syncthreads();
}

// dst_i = conj^cj(a_i) * (g0 + g1 u) for the Fq2 coefficients i >= first (others copied)
static __device__ __forceinline__ Fq29 fq2_scale_lane(const Fq29& x, const Fq29& y, const Fq29& g0, const Fq29& g1, int e) {
  // (x + y u)(g0 + g1 u) = (x g0 - y g1) + (x g1 + y g0) u ; every product N, so the lazy sum stays < 2^30
  return e ? fq29_add(fq29_mul(x, g1), fq29_mul(y, g0)) : fq29_sub(fq29_mul(x, g0), fq29_mul(y, g1));
}

// dst = a^(p^k), k in {1,2,3}: g_i -> conj^k(g_i) * gamma_{k,i}
static __device__ __noinline__ void coop_frob(int dst, int a, int k) {
  int tid = threadIdx.x;
  Fq29 out = fq29_zero();
  if (tid < 12) {
    int i = tid >> 1, e = tid & 1;
    Fq29 x = g_sh.r[a].v[2 * i], y = g_sh.r[a].v[2 * i + 1];
    if (k & 1) y = fq29_neg(y);
    if (i == 0) {
      out = e ? y : x;
    } else {
      Fq2_29 g = frob29_gamma(k, i);
      out = fq2_scale_lane(x, y, g.c0, g.c1, e);
    }
    out = fq29_norm(out);
  }
  __syncthreads();  // all reads of a done before dst (possibly == a) is written
  if (tid < 12) coop_store(dst, tid, out);
  __syncthreads();
}

// dst = a * (scal[0] + scal[1] u), an Fq2 scalar
static __device__ __noinline__ void coop_scale(int dst, int a) {
  int tid = threadIdx.x;
  Fq29 out = fq29_zero();
  if (tid < 12) {
    int i = tid >> 1, e = tid & 1;
    out = fq29_norm(fq2_scale_lane(g_sh.r[a].v[2 * i], g_sh.r[a].v[2 * i + 1], g_sh.scal[0], g_sh.scal[1], e));
  }
  __syncthreads();
  if (tid < 12) coop_store(dst, tid, out);
  __syncthreads();
}

// dst = a^-1 through the norms Fq12 -> Fq6 -> Fq2 -> Fq:
//   N = a conj(a) in Fq6;  adj = N^(p^2) N^(p^4);  d = N adj in Fq2;
//   a^-1 = conj(a) adj / d.      (uses RT0, RT1, RY0 as scratch; one Fq inversion chain on lane 0)
static __device__ __forceinline__ void coop_inv(int dst, int a) {
  coop_conj(RT0, a);                 // RT0 = conj(a)
  coop_mulr(RT1, a, RT0);         // RT1 = N
  coop_frob(RY0, RT1, 2);            // N^(p^2)
  coop_frob(dst, RY0, 2);            // N^(p^4)
  coop_mulr(RY0, RY0, dst);       // adj
  coop_mulr(RT1, RT1, RY0);       // d (Fq2: coefficients 0, 1)
  if (threadIdx.x == 0) {
    Fq29 d0 = g_sh.r[RT1].v[0], d1 = g_sh.r[RT1].v[1];
    Fq29 nrm = fq29_norm(fq29_add(fq29_sqr(d0), fq29_sqr(d1)));
    Fq29 ni = fq29_inv(fq29_canon_residue(nrm));
    g_sh.scal[0] = fq29_mul(d0, ni);
    g_sh.scal[1] = fq29_neg(fq29_mul(d1, ni));
  }
  __syncthreads();
  coop_mulr(dst, RT0, RY0);       // conj(a) adj
  coop_scale(dst, dst);              // / d
}

static __device__ __forceinline__ void coop_exp_by_x(int dst, int a) {
  // dst != a
  if (threadIdx.x < 12) g_sh.r[dst].v[threadIdx.x] = g_sh.r[a].v[threadIdx.x];
  __syncthreads();
  for (int i = 61; i >= 0; --i) {
    coop_mulr(dst, dst, dst);
    if ((BN254_X_U64 >> i) & 1ull) coop_mulr(dst, dst, a);
  }
}

#ifdef SNARKV_DECIDE_PROFILE  // dev aid: phase time stamps (100 MHz wall clock) overwrite the Gt output
#define DECIDE_STAMP(n) do { if (tid == 0 && gt_out) ((unsigned long long*)(gt_out + (size_t)i * 96))[n] = wall_clock64(); } while (0)
#else
#define DECIDE_STAMP(n) do { } while (0)
#endif

// The THROUGHPUT form (large decide_all batches): one 128-lane team per accumulator, 29 KB of LDS, four workgroups per CU.
__global__ void __launch_bounds__(kDecideThreads)
    k_decide(const G2Prepared29* __restrict__ prep, const uint32_t* __restrict__ accs, uint32_t m,
             uint8_t* __restrict__ ok, uint32_t* __restrict__ gt_out, uint32_t mont) {
  const int tid = threadIdx.x;
  const uint32_t i = blockIdx.x;
  if (i >= m) return;
  const uint32_t* a = accs + (size_t)i * 32;
  DECIDE_STAMP(0);
  if (tid < 4) {  // lhs.x, lhs.y, rhs.x, rhs.y -> Montgomery
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = a[8 * tid + j];
    g_sh.pt[tid >> 1][tid & 1] = fq29_canon_residue(fq29_from_words(w, mont != 0));
  }
  if (tid < 12) coop_store(RF, tid, tid == 0 ? fq29_one() : fq29_zero());
  if (tid < 12) {
    coop_store(RONE, tid, tid == 0 ? fq29_one() : fq29_zero());
    coop_store(RZERO, tid, fq29_zero());
  }
  __syncthreads();
  if (tid < 2)
    g_sh.live[tid] = !(fq29_limbs_all_zero(g_sh.pt[tid][0]) && fq29_limbs_all_zero(g_sh.pt[tid][1])) &&
                     !prep[tid].is_identity;
  // every line of both pairs evaluated at this accumulator's points, up front
  // and in parallel (2 x 102 x 4 products): l0 = cy*yP, l1 = cx*xP
  for (int j = tid; j < 2 * kLinesPerG2 * 4; j += kDecideThreads) {
    int k = j / (kLinesPerG2 * 4), rem = j % (kLinesPerG2 * 4), idx = rem / 4, c = rem % 4;
    g_lines4[k][idx][c] = fq29_mul(prep[k].line[idx].c[c], g_sh.pt[k][c < 2 ? 1 : 0]);
  }
  __syncthreads();

  DECIDE_STAMP(1);
  // ---- Miller loop (2 pairs, shared squarings); identity pairs contribute 1
  {
    int idx = 0;
    for (int b = kAteBits - 2; b >= 0; --b) {
      coop_mulr(RF, RF, RF);
      for (int k = 0; k < 2; ++k)
        if (g_sh.live[k]) coop_mull(k, idx, prep);
      ++idx;
      if (ate_bit(b)) {
        for (int k = 0; k < 2; ++k)
          if (g_sh.live[k]) coop_mull(k, idx, prep);
        ++idx;
      }
    }
    for (int s = 0; s < 2; ++s) {
      for (int k = 0; k < 2; ++k)
        if (g_sh.live[k]) coop_mull(k, idx, prep);
      ++idx;
    }
  }

  DECIDE_STAMP(2);
  // ---- final exponentiation, exact exponent (p^12-1)/r (see pairing.h)
  coop_inv(RINV, RF);
  DECIDE_STAMP(3);
  coop_conj(RT, RF);
  coop_mulr(RF, RT, RINV);          // f^(p^6-1)
  coop_frob(RT, RF, 2);
  coop_mulr(RF, RT, RF);            // ^(p^2+1)
  DECIDE_STAMP(4);
  coop_exp_by_x(RFX, RF);
  coop_exp_by_x(RFX2, RFX);
  coop_exp_by_x(RFX3, RFX2);
  DECIDE_STAMP(5);
  coop_frob(RY0, RF, 1);
  coop_frob(RT, RF, 2);
  coop_mulr(RY0, RY0, RT);
  coop_frob(RT, RF, 3);
  coop_mulr(RY0, RY0, RT);          // y0 = f^p f^(p^2) f^(p^3)
  coop_conj(RY1, RF);                  // y1 = 1/f
  coop_frob(RY2, RFX2, 2);             // y2
  coop_frob(RT, RFX, 1);
  coop_conj(RY3, RT);                  // y3
  coop_frob(RT, RFX2, 1);
  coop_mulr(RT, RFX, RT);
  coop_conj(RY4, RT);                  // y4
  coop_conj(RY5, RFX2);                // y5
  coop_frob(RT, RFX3, 1);
  coop_mulr(RT, RFX3, RT);
  coop_conj(RY6, RT);                  // y6
  coop_mulr(RT0, RY6, RY6);
  coop_mulr(RT0, RT0, RY4);
  coop_mulr(RT0, RT0, RY5);         // t0 = y6^2 y4 y5
  coop_mulr(RT1, RY3, RY5);
  coop_mulr(RT1, RT1, RT0);         // t1 = y3 y5 t0
  coop_mulr(RT0, RT0, RY2);         // t0 *= y2
  coop_mulr(RT1, RT1, RT1);
  coop_mulr(RT1, RT1, RT0);
  coop_mulr(RT1, RT1, RT1);         // t1 = (t1^2 t0)^2
  coop_mulr(RT0, RT1, RY1);         // t0 = t1 y1
  coop_mulr(RT1, RT1, RY0);         // t1 = t1 y0
  coop_mulr(RT0, RT0, RT0);
  coop_mulr(RF, RT0, RT1);          // result = t0^2 t1

  DECIDE_STAMP(6);
#ifdef SNARKV_DECIDE_PROFILE
  if (gt_out) return;
#endif
  // canonical words of the 12 coefficients (lane c), then the verdict
  __shared__ uint32_t canon[12][8];
  if (tid < 12) fq29_to_canonical(g_sh.r[RF].v[tid], canon[tid]);
  __syncthreads();
  if (tid == 0 && ok) {
    bool one = canon[0][0] == 1u;
    for (int c = 0; c < 12; ++c)
      for (int j = (c == 0 ? 1 : 0); j < 8; ++j) one = one && canon[c][j] == 0u;
    ok[i] = one ? 1 : 0;
  }
  if (gt_out && tid < 12) {
    // tower byte order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2  =  w^0, w^2, w^4, w^1, w^3, w^5
    const int wexp[6] = {0, 2, 4, 1, 3, 5};
    int pos = tid >> 1, e = tid & 1;
    uint32_t* dstw = gt_out + (size_t)i * 96 + (size_t)(2 * pos + e) * 8;
    for (int j = 0; j < 8; ++j) dstw[j] = canon[2 * wexp[pos] + e][j];
  }
}

// ------------------------------------------------------------------ D0: the line tables, one wavefront per key (g2_prepare_w.h)
// g2x2: g2 (128 B) || s_g2 (128 B), canonical or in-memory form; lanes 16 p + 2 t + e: point p, task t of the level, component e.
// Point 1 is NEGATED (the decider pairs rhs with -s_g2).  out[p]: the 29-bit table the decide kernels read.
__global__ void __launch_bounds__(64) k_g2_prepare_w(const uint32_t* __restrict__ g2x2, G2Prepared29* __restrict__ out,
                                                      const G2wTask* __restrict__ prog, uint32_t mont) {
  __shared__ Fq2_29P sl[2][kG2wSlots];
  __shared__ int ident[2];
  const int lane = threadIdx.x, pt = lane >> 4, t = (lane & 15) >> 1, e = lane & 1;
  const bool worker = pt < 2 && t < kG2wTasks;
  if (lane < 8) {  // the four coordinates of both points
    const int p = lane >> 2, c = lane & 3;  // c: x.c0, x.c1, y.c0, y.c1
    Fq29 v = fq29_canon_residue(fq29_from_words(g2x2 + 32 * p + 8 * c, mont != 0));
    if (p == 1 && c >= 2) v = fq29_canon_residue(fq29_neg(v));
    g2w_put(sl[p], c < 2 ? kG2wSlotQX : kG2wSlotQY, c & 1, v);
    g2w_put(sl[p], c < 2 ? kG2wSlotTX : kG2wSlotTYA, c & 1, v);
  } else if (lane < 8 + 24) {  // the constants of the program
    const int q = lane - 8, p = q / 12, k = (q % 12) >> 1, ee = q & 1;
    g2w_put(sl[p], k, ee, g2w_const(k, ee));  // slots 0 .. 5: ONE, B3, G12, G13, G22, G23
  } else if (lane < 8 + 24 + 8) {
    const int q = lane - 32, p = q >> 2, k = (q >> 1) & 1, ee = q & 1;
    g2w_put(sl[p], k ? kG2wSlotTZ : kG2wSlotTYB, ee, (k && !ee) ? fq29_one() : fq29_zero());  // Z = 1, YB = 0
  }
  __syncthreads();
  if (lane < 2) {
    ident[lane] = fq29_limbs_all_zero(g2w_get(sl[lane], kG2wSlotQX, 0)) && fq29_limbs_all_zero(g2w_get(sl[lane], kG2wSlotQX, 1)) &&
                  fq29_limbs_all_zero(g2w_get(sl[lane], kG2wSlotQY, 0)) && fq29_limbs_all_zero(g2w_get(sl[lane], kG2wSlotQY, 1));
    out[lane].is_identity = (uint32_t)ident[lane];
  }
  __syncthreads();
  G2wTask nxt;
  nxt.used = 0;
  if (worker) nxt = prog[t];
#pragma unroll 1
  for (int lv = 0; lv < kG2wLevels; ++lv) {
    Fq29 val = fq29_zero();
    const G2wTask tk = nxt;
    if (worker && lv + 1 < kG2wLevels) nxt = prog[(lv + 1) * kG2wTasks + t];  // fetched under this level's products
    const bool act = worker && tk.used && !ident[pt];
    if (act) {  // (a task's two lanes are neighbours and active together: the swap stays inside the pair)
      Fq29 am, bm, ap, bp;
      g2w_mine(sl[pt], tk, e, am, bm);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        ap.v[i] = (int32_t)dpp_u32<0xB1>((uint32_t)am.v[i]);  // quad_perm [1,0,3,2]
        bp.v[i] = (int32_t)dpp_u32<0xB1>((uint32_t)bm.v[i]);
      }
      val = g2w_product(am, bm, ap, bp, e);
    }
    __syncthreads();  // (one wavefront: no barrier instruction -- the loads of the level are done before its stores)
    if (act) {
      if (tk.dst >= 0) {
        g2w_put(sl[pt], tk.dst, e, val);
      } else {  // a line coefficient: canonical residue into the table
        const Fq29 cv = fq29_canon_of_product(val);
        out[pt].line[tk.out / 3].c[2 * (tk.out % 3) + e] = cv;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ D2: the program-driven latency form (decide_w.h)
// One workgroup of four wavefronts per accumulator; wavefronts (2 d, 2 d + 1) are duo d.  LDS: the value array of
// decide_w.h, then the program (two 8-byte operations per round).
__global__ void __launch_bounds__(256)
    k_decide_w(const G2Prepared29* __restrict__ prep, const uint32_t* __restrict__ accs, uint32_t m,
               uint8_t* __restrict__ ok, uint32_t* __restrict__ gt_out, const uint2* __restrict__ prog, int rounds, int result,
               uint32_t mont) {
  extern __shared__ Fq29P wt_lds[];
  __shared__ Fq29 pt[2][2];
  __shared__ int live[2];
  __shared__ uint32_t canon[12][8];
  uint2* prog_lds = reinterpret_cast<uint2*>(wt_lds + kWtValues);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, duo = wave >> 1, half = wave & 1;
  const uint32_t i = blockIdx.x;
  if (i >= m) return;
  const uint32_t* a = accs + (size_t)i * 32;
  if (tid < 4) {  // lhs.x, lhs.y, rhs.x, rhs.y -> Montgomery
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = a[8 * tid + j];
    pt[tid >> 1][tid & 1] = fq29_canon_residue(fq29_from_words(w, mont != 0));
  }
  for (int j = tid; j < 52; j += 256) wt_store(wt_lds, kWtConstBase + j, wt_const_value(j));
  for (int j = tid; j < 2 * rounds; j += 256) prog_lds[j] = prog[j];
  __syncthreads();
  if (tid < 2) live[tid] = !(fq29_limbs_all_zero(pt[tid][0]) && fq29_limbs_all_zero(pt[tid][1])) && !prep[tid].is_identity;
  __syncthreads();
  for (int t = tid; t < 2 * kLinesPerG2 * 3; t += 256) {
    const int pair = t / (kLinesPerG2 * 3), rem = t % (kLinesPerG2 * 3);
    wt_eval_line(wt_lds, prep, pair, rem / 3, rem % 3, pt[pair][0], pt[pair][1], live[pair] != 0);
  }
  __syncthreads();
  const WtLaneC LC = wt_lane_c(half, lane);
  uint2 raw = prog_lds[duo];
  for (int r = 0; r < rounds; ++r) {
    WtOp op;
    {
      const uint32_t w0 = __builtin_amdgcn_readfirstlane(raw.x), w1 = __builtin_amdgcn_readfirstlane(raw.y);
      op.dst = (uint16_t)w0, op.a = (uint16_t)(w0 >> 16), op.b = (uint16_t)w1, op.kind = (uint8_t)(w1 >> 16), op.flags = (uint8_t)(w1 >> 24);
    }
    if (r + 1 < rounds) raw = prog_lds[2 * (r + 1) + duo];  // the next round's operation, fetched under this one
    if (op.kind == WT_FQ2INV) {
      if (half == 0 && lane == 0) wt_fq2inv(wt_lds, op);
    } else if (op.kind != WT_IDLE) {
      const Fq29 own = wt_squeeze(group8_sum(wt_task_c(wt_lds, op, LC)));
      Fq29 other;  // the other u-component of the same power of w: 8 lanes away in the row
#pragma unroll
      for (int q = 0; q < 9; ++q) other.v[q] = (int32_t)dpp_u32<0x128>((uint32_t)own.v[q]);  // row_ror:8
      wt_write(wt_lds, op, half, lane, own, other);
    }
    __syncthreads();
  }
  if (tid < 12) fq29_to_canonical(wt_load(wt_lds, result + tid), canon[tid]);
  __syncthreads();
  if (tid == 0 && ok) {
    bool one = canon[0][0] == 1u;
    for (int c = 0; c < 12; ++c)
      for (int j = (c == 0 ? 1 : 0); j < 8; ++j) one = one && canon[c][j] == 0u;
    ok[i] = one ? 1 : 0;
  }
  if (gt_out && tid < 12) {
    // tower byte order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2  =  w^0, w^2, w^4, w^1, w^3, w^5
    const int wexp[6] = {0, 2, 4, 1, 3, 5};
    int pos = tid >> 1, e = tid & 1;
    uint32_t* dstw = gt_out + (size_t)i * 96 + (size_t)(2 * pos + e) * 8;
    for (int j = 0; j < 8; ++j) dstw[j] = canon[2 * wexp[pos] + e][j];
  }
}

// the program, built once per process and uploaded once per device
struct WtDeviceProgram {
  uint2* d_prog = nullptr;
  int rounds = 0, result = 0;
  size_t lds_bytes = 0;
};
static int wt_device_program(int device, const WtDeviceProgram** out) {
  static std::mutex mu;
  static WtProgram host;
  static bool built = false;
  static WtDeviceProgram dev[64];
  std::lock_guard<std::mutex> lk(mu);
  if (!built) {
    host = wt_build_program();
    built = true;
  }
  WtDeviceProgram& d = dev[device & 63];
  if (!d.d_prog) {
    static_assert(sizeof(WtOp) == sizeof(uint2), "an operation is one 8-byte word pair");
    std::vector<uint2> packed(host.ops.size());
    for (size_t q = 0; q < host.ops.size(); ++q) {
      const WtOp& o = host.ops[q];
      packed[q].x = (uint32_t)o.dst | ((uint32_t)o.a << 16);
      packed[q].y = (uint32_t)o.b | ((uint32_t)o.kind << 16) | ((uint32_t)o.flags << 24);
    }
    SNARKV_HIP(hipMalloc(&d.d_prog, packed.size() * sizeof(uint2)));
    SNARKV_HIP(hipMemcpy(d.d_prog, packed.data(), packed.size() * sizeof(uint2), hipMemcpyHostToDevice));
    d.rounds = host.rounds;
    d.result = host.result;
    d.lds_bytes = kWtLdsBytes + packed.size() * sizeof(uint2);
    SNARKV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_decide_w), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)d.lds_bytes));
  }
  *out = &d;
  return SNARKV_OK;
}

int launch_g2_prepare(snarkv_ctx* ctx, const void* d_g2x2_256, void* d_prep) {
  static std::mutex mu;
  static G2wTask* d_prog[64] = {nullptr};
  G2wTask* prog = nullptr;
  {
    std::lock_guard<std::mutex> lk(mu);
    G2wTask*& slot = d_prog[ctx->device & 63];
    if (!slot) {  // the level program: uploaded once per device
      SNARKV_HIP(hipMalloc(&slot, sizeof(kG2wProg)));
      SNARKV_HIP(hipMemcpy(slot, kG2wProg, sizeof(kG2wProg), hipMemcpyHostToDevice));
    }
    prog = slot;
  }
  G2Prepared29* d29 = reinterpret_cast<G2Prepared29*>((char*)d_prep + prep29_offset());
  hipLaunchKernelGGL(k_g2_prepare_w, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)d_g2x2_256, d29, (const G2wTask*)prog,
                     ctx->mont ? 1u : 0u);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

int launch_validate_g2(snarkv_ctx* ctx, const void* d_g2x2_256, int* bad_host) {
  void* d_bad = nullptr;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_FLAGS, 64, &d_bad));
  SNARKV_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_validate_g2, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t*)d_g2x2_256, (int*)d_bad,
                     ctx->mont ? 1u : 0u);
  SNARKV_HIP(hipGetLastError());
  SNARKV_HIP(hipMemcpyAsync(bad_host, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  return SNARKV_OK;
}

// Which form: batches of up to 256 accumulators take the program-driven latency kernel (one workgroup of four
// wavefronts + ~95 KiB of LDS per accumulator: one per CU), larger ones the one-team throughput kernel (two wavefronts,
// 29 KiB, four workgroups per CU).  SNARKV_DECIDE_FORM = 1 (one team) / 3 (program) forces one (test / A-B knob); both give
// the same bits.  (The round-3 two-team latency form, 0.87 ms per decide against 0.56: profiles/r04_ab_decide_wave.txt.)
int launch_decide(snarkv_ctx* ctx, const void* d_prep, const void* d_accs, size_t m, void* d_ok, void* d_gt) {
  const G2Prepared29* d29 = reinterpret_cast<const G2Prepared29*>((const char*)d_prep + prep29_offset());
  const uint32_t mont = ctx->mont ? 1u : 0u;  // the accumulators' points in halo2curves' in-memory form
  int form = m <= 256 ? 3 : 1;
  if (const char* e = getenv("SNARKV_DECIDE_FORM")) {
    int v = atoi(e);
    if (v == 1 || v == 3) form = v;
  }
  if (form == 3) {
    const WtDeviceProgram* wp = nullptr;
    SNARKV_TRY(wt_device_program(ctx->device, &wp));
    hipLaunchKernelGGL(k_decide_w, dim3((uint32_t)m), dim3(256), wp->lds_bytes, ctx->stream, d29, (const uint32_t*)d_accs,
                       (uint32_t)m, (uint8_t*)d_ok, (uint32_t*)d_gt, (const uint2*)wp->d_prog, wp->rounds, wp->result, mont);
  } else
    hipLaunchKernelGGL(k_decide, dim3((uint32_t)m), dim3(kDecideThreads), 0, ctx->stream, d29,
                       (const uint32_t*)d_accs, (uint32_t)m, (uint8_t*)d_ok, (uint32_t*)d_gt, mont);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
