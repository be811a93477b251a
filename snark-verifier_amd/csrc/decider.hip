// KZG pairing decider on the device.
// Replaces the native `AccumulationDecider` impl for `KzgAs`
// (reference snark-verifier/src/pcs/kzg/decider.rs:70-93):
//   decide     : e(lhs, g2) * e(rhs, -s_g2) == 1
//   decide_all : the same for every accumulator of a list (decider.rs:84-93)
//
//   D0 k_g2_prepare : once per deciding key -- line tables of g2 and -s_g2
//                     (the reference's `G2Prepared::from`, redone there on
//                     every call, decider.rs:74)
//   D1 k_decide     : ONE 256-lane WORKGROUP per accumulator (pairing_coop.cuh):
//                     2-pair Miller loop with shared squarings + exact final
//                     exponentiation + `is_identity`; every Fq12 product is one
//                     parallel round of 204 Fq products + a two-stage LDS sum.
//                     (v1 gave one lane per accumulator: 45-55 ms per decide,
//                     whatever the batch size; this form: ~2 ms.)
// Batches of independent accumulators (decide_all) are the second parallel axis
// (SURVEY.md 8e).
#include "ctx.hpp"
#include "g1.cuh"
#include "pairing.cuh"
#include "pairing_coop.cuh"

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

// ------------------------------------------------------------------ D1
struct CoopReg {  // an Fq12 in the flat basis, plus 9x (operand form for the w^6 wrap)
  Fq v[12];
  Fq v9[12];
};

enum { RF = 0, RT, RINV, RFX, RFX2, RFX3, RY0, RY1, RY2, RY3, RY4, RY5, RY6, RT0, RT1, RCOUNT };

struct CoopShared {
  CoopReg r[RCOUNT];
  Fq prods[COOP_NPROD];
  Fq parts[48];
  Fq lines[2][kLinesPerG2][6];  // per pair, per line: l0 (2), l1 (2), l2 (2) already times yP / xP
  G1AffineM pt[2];
  int live[2];
};

// One workgroup = one accumulator; the whole state lives in LDS (file-scope so
// that every helper addresses it with ds_* instructions, not flat pointers).
__shared__ CoopShared g_sh;

// Per-lane descriptors, loaded once into registers (never re-read from memory).
struct CoopLane {
  uint32_t prod;      // s | t<<4 | use9<<8, or 0xFFFF
  uint32_t st1[3];    // six 16-bit stage-1 entries
};

static __device__ __forceinline__ CoopLane coop_lane_init() {
  CoopLane L;
  int tid = threadIdx.x;
  L.prod = tid < COOP_NPROD ? kCoopProd[tid] : 0xFFFFu;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    uint32_t lo = tid < 48 ? kCoopStage1[tid][2 * k] : 0xFFFFu;
    uint32_t hi = tid < 48 ? kCoopStage1[tid][2 * k + 1] : 0xFFFFu;
    L.st1[k] = lo | (hi << 16);
  }
  return L;
}

// b operand of a product round: a register of g_sh.r, or a sparse line
struct CoopB {
  int reg;   // >= 0: g_sh.r[reg].v
  int pair;  // line operand: g_sh.lines[pair][idx]
  int idx;
};

// flat slot t of a sparse line (non-zero slots 0,1,2,3,6,7 -> 0..5), -1 = zero
static __device__ __forceinline__ int line_slot(unsigned t) { return t < 4 ? (int)t : (t == 6 ? 4 : (t == 7 ? 5 : -1)); }

// dst = a * b.  All 256 lanes call it.
static __device__ __forceinline__ void coop_mul_b(const CoopLane& L, int dst, int a, CoopB b) {
  int tid = threadIdx.x;
  if (tid < COOP_NPROD) {
    unsigned s = L.prod & 15u, t = (L.prod >> 4) & 15u;
    const Fq& x = (L.prod >> 8) ? g_sh.r[a].v9[s] : g_sh.r[a].v[s];
    if (b.reg >= 0) {
      g_sh.prods[tid] = fq_mul(x, g_sh.r[b.reg].v[t]);
    } else {
      int sl = line_slot(t);
      g_sh.prods[tid] = sl >= 0 ? fq_mul(x, g_sh.lines[b.pair][b.idx][sl]) : fq_zero();
    }
  }
  __syncthreads();
  if (tid < 48) {
    Fq acc = fq_zero();
#pragma unroll
    for (int k = 0; k < COOP_STAGE1_TERMS; ++k) {
      unsigned e = (L.st1[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
      if (e != 0xFFFFu) {
        const Fq& pr = g_sh.prods[e & 0x7FFFu];
        acc = (e & 0x8000u) ? fq_sub(acc, pr) : fq_add(acc, pr);
      }
    }
    g_sh.parts[tid] = acc;
  }
  __syncthreads();
  if (tid < 12) {
    Fq c = fq_add(fq_add(g_sh.parts[4 * tid], g_sh.parts[4 * tid + 1]),
                  fq_add(g_sh.parts[4 * tid + 2], g_sh.parts[4 * tid + 3]));
    g_sh.r[dst].v[tid] = c;
    g_sh.r[dst].v9[tid] = fq_mul9(c);
  }
  __syncthreads();
}

static __device__ __noinline__ void coop_mulr(const CoopLane& L, int dst, int a, int b) {
  coop_mul_b(L, dst, a, CoopB{b, 0, 0});
}
static __device__ __noinline__ void coop_mull(const CoopLane& L, int pair, int idx) {
  coop_mul_b(L, RF, RF, CoopB{-1, pair, idx});
}

// dst = conj(a): negate the odd powers of w (the c1 half of the tower)
static __device__ __noinline__ void coop_conj(int dst, int a) {
  int tid = threadIdx.x;
  if (tid < 12) {
    bool odd = ((tid >> 1) & 1) != 0;
    Fq x = g_sh.r[a].v[tid], x9 = g_sh.r[a].v9[tid];
    g_sh.r[dst].v[tid] = odd ? fq_neg(x) : x;
    g_sh.r[dst].v9[tid] = odd ? fq_neg(x9) : x9;
  }
  __syncthreads();
}

// dst = a^(p^k), k in {1,2,3}: g_i -> conj^k(g_i) * gamma_{k,i}
static __device__ __noinline__ void coop_frob(int dst, int a, int k) {
  int tid = threadIdx.x;
  Fq out = fq_zero();
  if (tid < 12) {
    int i = tid >> 1, e = tid & 1;
    Fq x = g_sh.r[a].v[2 * i], y = g_sh.r[a].v[2 * i + 1];
    if (k & 1) y = fq_neg(y);
    if (i == 0) {
      out = e ? y : x;
    } else {
      Fq2 g = frob_gamma(k, i);
      out = e ? fq_add(fq_mul(x, g.c1), fq_mul(y, g.c0)) : fq_sub(fq_mul(x, g.c0), fq_mul(y, g.c1));
    }
  }
  __syncthreads();  // all reads of a done before dst (possibly == a) is written
  if (tid < 12) {
    g_sh.r[dst].v[tid] = out;
    g_sh.r[dst].v9[tid] = fq_mul9(out);
  }
  __syncthreads();
}

// dst = a^-1: tower inversion on lane 0 (one Fq inversion chain dominates it)
static __device__ __noinline__ void coop_inv(int dst, int a) {
  if (threadIdx.x == 0) {
    Fq flat[12];
    for (int c = 0; c < 12; ++c) flat[c] = g_sh.r[a].v[c];
    Fq12 ti = fq12_inv(coop_tower_from_flat(flat));
    coop_flat_from_tower(ti, flat);
    for (int c = 0; c < 12; ++c) g_sh.r[dst].v[c] = flat[c];
  }
  __syncthreads();
  if (threadIdx.x < 12) g_sh.r[dst].v9[threadIdx.x] = fq_mul9(g_sh.r[dst].v[threadIdx.x]);
  __syncthreads();
}

static __device__ __noinline__ void coop_exp_by_x(const CoopLane& L, int dst, int a) {
  // dst != a
  if (threadIdx.x < 12) {
    g_sh.r[dst].v[threadIdx.x] = g_sh.r[a].v[threadIdx.x];
    g_sh.r[dst].v9[threadIdx.x] = g_sh.r[a].v9[threadIdx.x];
  }
  __syncthreads();
  for (int i = 61; i >= 0; --i) {
    coop_mulr(L, dst, dst, dst);
    if ((BN254_X_U64 >> i) & 1ull) coop_mulr(L, dst, dst, a);
  }
}

__global__ void __launch_bounds__(256)
    k_decide(const G2Prepared* __restrict__ prep, const uint32_t* __restrict__ accs, uint32_t m,
             uint8_t* __restrict__ ok, uint32_t* __restrict__ gt_out) {
  const int tid = threadIdx.x;
  const uint32_t i = blockIdx.x;
  if (i >= m) return;
  const CoopLane L = coop_lane_init();
  const uint32_t* a = accs + (size_t)i * 32;
  if (tid < 4) {  // lhs.x, lhs.y, rhs.x, rhs.y -> Montgomery
    Fq v = load_fq_canonical(a + 8 * tid);
    if (tid == 0) g_sh.pt[0].x = v;
    if (tid == 1) g_sh.pt[0].y = v;
    if (tid == 2) g_sh.pt[1].x = v;
    if (tid == 3) g_sh.pt[1].y = v;
  }
  if (tid < 12) {
    Fq one = fq_one();
    g_sh.r[RF].v[tid] = tid == 0 ? one : fq_zero();
    g_sh.r[RF].v9[tid] = tid == 0 ? fq_mul9(one) : fq_zero();
  }
  __syncthreads();
  if (tid < 2)
    g_sh.live[tid] = !(fq_is_zero(g_sh.pt[tid].x) && fq_is_zero(g_sh.pt[tid].y)) && !prep[tid].is_identity;
  // every line of both pairs evaluated at this accumulator's points, up front
  // and in parallel (2 x 102 x 6 coefficients): l0 = cy*yP, l1 = cx*xP, l2 = cw
  for (int j = tid; j < 2 * kLinesPerG2 * 6; j += 256) {
    int k = j / (kLinesPerG2 * 6), rem = j % (kLinesPerG2 * 6), idx = rem / 6, c = rem % 6;
    const LineCoeff& l = prep[k].line[idx];
    Fq v;
    if (c == 0) v = fq_mul(l.cy.c0, g_sh.pt[k].y);
    else if (c == 1) v = fq_mul(l.cy.c1, g_sh.pt[k].y);
    else if (c == 2) v = fq_mul(l.cx.c0, g_sh.pt[k].x);
    else if (c == 3) v = fq_mul(l.cx.c1, g_sh.pt[k].x);
    else if (c == 4) v = l.cw.c0;
    else v = l.cw.c1;
    g_sh.lines[k][idx][c] = v;
  }
  __syncthreads();

  // ---- Miller loop (2 pairs, shared squarings); identity pairs contribute 1
  int idx = 0;
  for (int b = kAteBits - 2; b >= 0; --b) {
    coop_mulr(L, RF, RF, RF);
    for (int k = 0; k < 2; ++k)
      if (g_sh.live[k]) coop_mull(L, k, idx);
    ++idx;
    if (ate_bit(b)) {
      for (int k = 0; k < 2; ++k)
        if (g_sh.live[k]) coop_mull(L, k, idx);
      ++idx;
    }
  }
  for (int s = 0; s < 2; ++s) {
    for (int k = 0; k < 2; ++k)
      if (g_sh.live[k]) coop_mull(L, k, idx);
    ++idx;
  }

  // ---- final exponentiation, exact exponent (p^12-1)/r (see pairing.cuh)
  coop_conj(RT, RF);
  coop_inv(RINV, RF);
  coop_mulr(L, RF, RT, RINV);          // f^(p^6-1)
  coop_frob(RT, RF, 2);
  coop_mulr(L, RF, RT, RF);            // ^(p^2+1)
  coop_exp_by_x(L, RFX, RF);
  coop_exp_by_x(L, RFX2, RFX);
  coop_exp_by_x(L, RFX3, RFX2);
  coop_frob(RY0, RF, 1);
  coop_frob(RT, RF, 2);
  coop_mulr(L, RY0, RY0, RT);
  coop_frob(RT, RF, 3);
  coop_mulr(L, RY0, RY0, RT);          // y0 = f^p f^(p^2) f^(p^3)
  coop_conj(RY1, RF);                  // y1 = 1/f
  coop_frob(RY2, RFX2, 2);             // y2
  coop_frob(RT, RFX, 1);
  coop_conj(RY3, RT);                  // y3
  coop_frob(RT, RFX2, 1);
  coop_mulr(L, RT, RFX, RT);
  coop_conj(RY4, RT);                  // y4
  coop_conj(RY5, RFX2);                // y5
  coop_frob(RT, RFX3, 1);
  coop_mulr(L, RT, RFX3, RT);
  coop_conj(RY6, RT);                  // y6
  coop_mulr(L, RT0, RY6, RY6);
  coop_mulr(L, RT0, RT0, RY4);
  coop_mulr(L, RT0, RT0, RY5);         // t0 = y6^2 y4 y5
  coop_mulr(L, RT1, RY3, RY5);
  coop_mulr(L, RT1, RT1, RT0);         // t1 = y3 y5 t0
  coop_mulr(L, RT0, RT0, RY2);         // t0 *= y2
  coop_mulr(L, RT1, RT1, RT1);
  coop_mulr(L, RT1, RT1, RT0);
  coop_mulr(L, RT1, RT1, RT1);         // t1 = (t1^2 t0)^2
  coop_mulr(L, RT0, RT1, RY1);         // t0 = t1 y1
  coop_mulr(L, RT1, RT1, RY0);         // t1 = t1 y0
  coop_mulr(L, RT0, RT0, RT0);
  coop_mulr(L, RF, RT0, RT1);          // result = t0^2 t1

  if (tid == 0 && ok) {
    bool one = fq_eq(g_sh.r[RF].v[0], fq_one());
    for (int c = 1; c < 12; ++c) one = one && fq_is_zero(g_sh.r[RF].v[c]);
    ok[i] = one ? 1 : 0;
  }
  if (gt_out && tid < 12) {
    // tower byte order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2  =  w^0, w^2, w^4, w^1, w^3, w^5
    const int wexp[6] = {0, 2, 4, 1, 3, 5};
    int pos = tid >> 1, e = tid & 1;
    store_fq_canonical(g_sh.r[RF].v[2 * wexp[pos] + e], gt_out + (size_t)i * 96 + (size_t)(2 * pos + e) * 8);
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
  hipLaunchKernelGGL(k_decide, dim3((uint32_t)m), dim3(256), 0, ctx->stream, (const G2Prepared*)d_prep,
                     (const uint32_t*)d_accs, (uint32_t)m, (uint8_t*)d_ok, (uint32_t*)d_gt);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
