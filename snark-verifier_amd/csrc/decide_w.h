// The latency form of the KZG pairing decider: ONE workgroup of four wavefronts per accumulator runs a STATIC PROGRAM of
// Fq12 operations (decide_sched.hpp builds it on the host, once) -- reference snark-verifier/src/pcs/kzg/decider.rs:70-82,
// i.e. halo2curves' `multi_miller_loop` + `final_exponentiation` + `is_identity`.
//
// Why a program.  One decide is a dependency chain of ~350 Fq12 products; the round-3 kernel spent 2.2 us per product,
// two thirds of it outside the multiplier (operand addressing, 36 single-word LDS reads, separate low / high sums with a
// 64-bit finalisation, two barriers, call overhead), and hand-wrote its overlap of independent work (line products next
// to the Miller squarings, R *= S next to S <- S^2).  Here
//   * a DUO (two wavefronts on two SIMDs) runs one product per round: wavefront h owns output powers w^(3h) .. w^(3h+2),
//     6 fused two-product Montgomery steps per output coefficient in an aligned 8-lane group, so a group's DPP butterfly
//     holds the whole coefficient -- the xi-wrap (w^6 = xi = 9 + u) is NOT applied at the sum: every value lives in LDS
//     twice, as c and as xi * c, and a wrapped pair simply reads the second copy.  One 32-bit sum, a fused
//     squeeze (float-estimated quotient, one carry pass), 9 DPP moves for the partner component, one xi pass, stores;
//   * the two duos of a workgroup run two independent operations per round; the host-side list scheduler places every
//     operation of the whole decide (Miller loop: f <- f^2, f <- f * (l1 l2 [l1' l2']) with the pair products computed
//     ahead; final exponentiation: right-to-left exponentiations by x with R *= S beside S <- S^2, the hard part's
//     independent Frobenius maps) on that 2 x N grid by critical path, allocates LDS registers by liveness, and the
//     kernel is a loop over rounds with ONE barrier each (an operation never writes a register that is read in its round).
// Operands are addressed by per-lane offsets computed once; values are 40-byte records (8-byte aligned: ds_read2_b64).
//
// Everything a lane computes is in this header and compiles for the host: tests/hosttest runs the whole program lane by
// lane (DPP exchanges replaced by explicit sums) against the tower arithmetic and the oracle's Gt bytes.
#pragma once
#include "pairing.h"
#include "pairing_coop29.h"

namespace snarkv {

// line table in the lazy 29-bit form the decide kernels consume:
// c[0..5] = cy.c0, cy.c1, cx.c0, cx.c1, cw.c0, cw.c1 (canonical residues, Montgomery R = 2^261)
struct LineCoeff29 {
  Fq29 c[6];
};
struct G2Prepared29 {
  LineCoeff29 line[kLinesPerG2];
  uint32_t is_identity;
  uint32_t pad[3];
};

struct alignas(8) Fq29P {  // one value in LDS: 9 limbs + a pad word
  int32_t v[10];
};

// ---- the workgroup's LDS, in units of Fq29P -------------------------------------------------------------------------
constexpr int kWtDense = 24;   // a dense register: [c] = coefficient c = 2 i + e  <->  u^e w^i ;  [12 + c] = xi * that
constexpr int kWtRegs = 16;    // dense registers the scheduler may use (it needs 10 with a look-ahead of five Miller steps)
constexpr int kWtConstBase = kWtRegs * kWtDense;
// pseudo-registers (plain halves only): Frobenius constants gamma_{k,i} for k = 1, 2, 3 (i = 0: one), the all-ones
// register, and the Fq2 scalar register the inversion broadcasts
constexpr int kWtGamma0 = kWtConstBase;            // + 12 (k - 1)
constexpr int kWtOnes = kWtConstBase + 36;
constexpr int kWtScalar = kWtConstBase + 48;       // 2 values
constexpr int kWtLineA0 = kWtConstBase + 52;       // pair 0 (A role): 6 values per line  [2 s + e], s = slot of w^0, w^1, w^3
constexpr int kWtLineB0 = kWtLineA0 + 6 * kLinesPerG2;  // pair 1 (B role): 12 values per line: plain, then xi *
constexpr int kWtValues = kWtLineB0 + 12 * kLinesPerG2;
constexpr size_t kWtLdsBytes = (size_t)kWtValues * sizeof(Fq29P);

// ---- one operation of a duo ---------------------------------------------------------------------------------------
enum : uint8_t { WT_IDLE = 0, WT_MUL = 1, WT_PW = 2, WT_FQ2INV = 3 };
enum : uint8_t {
  WT_A_LINE = 1,    // A is a pair-0 line (3 coefficients: w^0, w^1, w^3)
  WT_B_LINE = 2,    // B is a pair-1 line
  WT_A_CONJ = 4,    // use conj(A): the odd powers of w negated
  WT_B_CONJ = 8,
  WT_A_UCONJ = 16,  // PW: use the Fq2-conjugate of every coefficient of A (odd Frobenius maps)
  WT_B_BCAST = 32,  // PW: B's coefficient 0 for every output (multiplication by an Fq2 scalar)
};
struct WtOp {
  uint16_t dst, a, b;  // value indices (register bases, line bases)
  uint8_t kind, flags;
};
static_assert(sizeof(WtOp) == 8, "two operations of a round = one 16-byte load");

// ---- a lane's task -----------------------------------------------------------------------------------------------------
// Wavefront half h, lane l = 8 g + jj: output coefficient (k, e) = (3 h + (g >> 1), g & 1), g < 6.
//   MUL: jj = j < 6 is the power of w taken from A; B gives power i2 = (k - j) mod 6, from the xi copy when j > k.
//   PW : only jj = 0 works: A's coefficient k times B's coefficient k (or 0).
struct WtLane {
  int k, e, j, i2;
  bool group, xi;  // group: g < 6 (the lane belongs to an output);  xi: the pair wraps
};
SNARKV_HD WtLane wt_lane(int half, int lane) {
  WtLane L;
  const int g = lane >> 3;
  L.group = g < 6;
  L.k = 3 * half + ((g >> 1) % 3);
  L.e = g & 1;
  L.j = lane & 7;
  L.i2 = (L.k - L.j + 12) % 6;
  L.xi = L.j > L.k;
  return L;
}
SNARKV_HD int wt_line_slot(int i) { return i == 3 ? 2 : i; }  // w^0, w^1, w^3 -> 0, 1, 2
SNARKV_HD bool wt_line_has(int i) { return i == 0 || i == 1 || i == 3; }

SNARKV_HD Fq29 wt_load(const Fq29P* lds, int idx) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = lds[idx].v[i];
  return r;
}
SNARKV_HD void wt_store(Fq29P* lds, int idx, const Fq29& x) {
#pragma unroll
  for (int i = 0; i < 9; ++i) lds[idx].v[i] = x.v[i];
}

// x or -x, limb-wise (m = 0 / -1)
SNARKV_HD Fq29 wt_cneg(const Fq29& x, int32_t m) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = (x.v[i] ^ m) - m;
  return r;
}

// What a lane needs of an operation, as far as it does not depend on the operation: computed once per kernel.
// Offsets are in values from the operand's base; -1: the lane has no task in that addressing mode.
struct WtLaneC {
  int a_dense, a_line, a_pw;         // A's coefficient (u^0 component; u^1 follows it)
  int b_dense, b_line, b_pw, b_pwb;  // B's component y0; y1 lies at y0 + dy
  int dy;                            // +1 (e = 0) / -1 (e = 1)
  int32_t m1;                        // sign of the second term: -1 (e = 0: a0 y0 - a1 y1) / 0
  int32_t odd_a, odd_b, odd_k;       // -1 where the power of w taken from A / from B / of the output is odd (conj flags)
};
SNARKV_HD WtLaneC wt_lane_c(int half, int lane) {
  const WtLane L = wt_lane(half, lane);
  WtLaneC C;
  const bool mul = L.group && L.j < 6, pw = L.group && L.j == 0;
  C.a_dense = mul ? 2 * L.j : -1;
  C.a_line = mul && wt_line_has(L.j) ? 2 * wt_line_slot(L.j) : -1;
  C.a_pw = pw ? 2 * L.k : -1;
  C.b_dense = mul ? (L.xi ? 12 : 0) + 2 * L.i2 + L.e : -1;
  C.b_line = mul && wt_line_has(L.i2) ? (L.xi ? 6 : 0) + 2 * wt_line_slot(L.i2) + L.e : -1;
  C.b_pw = pw ? 2 * L.k + L.e : -1;
  C.b_pwb = pw ? L.e : -1;
  C.dy = L.e ? -1 : 1;
  C.m1 = L.e ? 0 : -1;
  C.odd_a = (L.j & 1) ? -1 : 0;
  C.odd_b = (L.i2 & 1) ? -1 : 0;
  C.odd_k = (L.k & 1) ? -1 : 0;
  return C;
}

// The lane's contribution to its output coefficient: one fused two-product Montgomery step, or zero.
//   e = 0:  a0 y0 - a1 y1      e = 1:  a0 y0 + a1 y1       with (y0, y1) = (b_e, b_(1-e)) of the chosen B coefficient
// Every operand is carry-normalised (|limb| < 2^29): plain values within 1.5 p, xi copies within 15 p, so the sum of the
// two products is below 45 p^2 and the step's output within (-0.3 p, 1.3 p).
SNARKV_HD Fq29 wt_task_c(const Fq29P* lds, const WtOp op, const WtLaneC& C) {
  const bool pw = op.kind == WT_PW;
  // the offset of the operation's addressing mode, picked with uniform masks (an indexed pick would put the lane's
  // constants on the stack: a scratch load at the head of every round)
  const int m_pw = pw ? -1 : 0, m_al = (!pw && (op.flags & WT_A_LINE)) ? -1 : 0, m_ad = ~(m_pw | m_al);
  const int m_pb = (pw && (op.flags & WT_B_BCAST)) ? -1 : 0, m_pp = m_pw & ~m_pb;
  const int m_bl = (!pw && (op.flags & WT_B_LINE)) ? -1 : 0, m_bd = ~(m_pw | m_bl);
  const int ia = (C.a_pw & m_pw) | (C.a_line & m_al) | (C.a_dense & m_ad);
  const int ib = (C.b_pwb & m_pb) | (C.b_pw & m_pp) | (C.b_line & m_bl) | (C.b_dense & m_bd);
  if (ia < 0 || ib < 0 || (op.kind != WT_MUL && op.kind != WT_PW)) return fq29_zero();
  Fq29 a0 = wt_load(lds, op.a + ia), a1 = wt_load(lds, op.a + ia + 1);
  const Fq29 y0 = wt_load(lds, op.b + ib), y1 = wt_load(lds, op.b + ib + C.dy);
  int32_t m1 = C.m1;
  if (op.flags & (WT_A_CONJ | WT_B_CONJ | WT_A_UCONJ)) {  // (uniform; a fifth of the hard part's operations)
    int32_t m0 = 0;
    if (op.flags & WT_A_CONJ) m0 ^= pw ? C.odd_k : C.odd_a;
    if (op.flags & WT_B_CONJ) m0 ^= C.odd_b;
    m1 ^= m0;
    if (op.flags & WT_A_UCONJ) m1 = ~m1;
    a0 = wt_cneg(a0, m0);
  }
  return fq29_mul2(a0, y0, wt_cneg(a1, m1), y1);
}
SNARKV_HD Fq29 wt_task(const Fq29P* lds, const WtOp op, int half, int lane) { return wt_task_c(lds, op, wt_lane_c(half, lane)); }

// s = limb-wise sum of <= 6 task outputs (limbs 0..7 read as unsigned: < 6 * 2^29; limb 8 signed).  Returns the
// carry-normalised representative within 0.5 p (+ the estimate's slack: < 1.5 p) of the same residue: the quotient is
// estimated from the top limb alone (the lower limbs carry at most 6 into it, against p >> 232 = 2^21.6).
SNARKV_HD Fq29 wt_squeeze(const Fq29& s) {
  const float inv_ptop = 1.0f / (float)fq29_p(8);
  const float qf = (float)s.v[8] * inv_ptop;
  const int32_t q = (int32_t)(qf + (qf >= 0 ? 0.5f : -0.5f));
  Fq29 r;
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t t = (int64_t)(uint32_t)s.v[i] - (int64_t)q * fq29_p(i) + c;
    r.v[i] = (int32_t)t & kMask29;
    c = t >> 29;
  }
  r.v[8] = (int32_t)((int64_t)s.v[8] - (int64_t)q * fq29_p(8) + c);
  return r;
}

// component e of xi * (c0 + c1 u) = (9 c0 - c1) + (c0 + 9 c1) u, from own = c_e and other = c_(1-e); carry-normalised,
// within 15 p for inputs within 1.5 p
SNARKV_HD Fq29 wt_xi(const Fq29& own, const Fq29& other, int e) {
  Fq29 r;
  int64_t c = 0;
  const int32_t m = e ? 0 : -1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t t = 9 * (int64_t)own.v[i] + (int64_t)((other.v[i] ^ m) - m) + c;
    r.v[i] = (int32_t)t & kMask29;
    c = t >> 29;
  }
  r.v[8] = (int32_t)(9 * (int64_t)own.v[8] + (int64_t)((other.v[8] ^ m) - m) + c);
  return r;
}

// the writer lane (jj = 0 of a group) stores its coefficient and the xi copy
SNARKV_HD void wt_write(Fq29P* lds, const WtOp op, int half, int lane, const Fq29& own, const Fq29& other) {
  const WtLane L = wt_lane(half, lane);
  if (!L.group || L.j != 0 || (op.kind != WT_MUL && op.kind != WT_PW)) return;
  const int c = 2 * L.k + L.e;
  wt_store(lds, op.dst + c, own);
  wt_store(lds, op.dst + 12 + c, wt_xi(own, other, L.e));
}

// WT_FQ2INV, one lane: a's coefficient 0 is d = d0 + d1 u (the Fq2 norm at the bottom of the Fq12 inversion);
// the scalar register <- 1 / d = (d0 - d1 u) / (d0^2 + d1^2)
SNARKV_HD void wt_fq2inv(Fq29P* lds, const WtOp op) {
  const Fq29 d0 = wt_load(lds, op.a), d1 = wt_load(lds, op.a + 1);
  const Fq29 nrm = fq29_norm(fq29_add(fq29_sqr(d0), fq29_sqr(d1)));
  const Fq29 ni = fq29_inv(fq29_canon_residue(nrm));
  wt_store(lds, op.dst, fq29_mul(d0, ni));
  wt_store(lds, op.dst + 1, fq29_norm(fq29_neg(fq29_mul(d1, ni))));
}

// ---- set-up pieces (lane-parallel, any lane count) ----------------------------------------------------------------------
// constants: value j of the 52 behind kWtConstBase
SNARKV_HD Fq29 wt_const_value(int j) {
  if (j < 36) {  // gamma_{k, i}: register k - 1, coefficient 2 i + e
    const int k = j / 12 + 1, c = j % 12, i = c >> 1, e = c & 1;
    if (i == 0) return e ? fq29_zero() : fq29_one();
    const Fq2_29 g = frob29_gamma(k, i);
    return e ? g.c1 : g.c0;
  }
  if (j < 48) return (j & 1) ? fq29_zero() : fq29_one();  // ones
  return fq29_zero();                                      // scalar register
}

// line (pair, idx), slot s (w^0: cy * yP, w^1: cx * xP, w^3: cw), both components + (pair 1) the xi copies.
// A dead pair (its G1 or G2 point is the identity) contributes the constant 1.
SNARKV_HD void wt_eval_line(Fq29P* lds, const G2Prepared29* prep, int pair, int idx, int s, const Fq29& px, const Fq29& py,
                            bool live) {
  Fq29 v0, v1;
  if (!live) {
    v0 = s == 0 ? fq29_one() : fq29_zero();
    v1 = fq29_zero();
  } else {
    const LineCoeff29& l = prep[pair].line[idx];
    if (s == 0) {
      v0 = fq29_mul(l.c[0], py);
      v1 = fq29_mul(l.c[1], py);
    } else if (s == 1) {
      v0 = fq29_mul(l.c[2], px);
      v1 = fq29_mul(l.c[3], px);
    } else {
      v0 = l.c[4];
      v1 = l.c[5];
    }
  }
  if (pair == 0) {
    wt_store(lds, kWtLineA0 + 6 * idx + 2 * s, v0);
    wt_store(lds, kWtLineA0 + 6 * idx + 2 * s + 1, v1);
  } else {
    const int b = kWtLineB0 + 12 * idx + 2 * s;
    wt_store(lds, b, v0);
    wt_store(lds, b + 1, v1);
    wt_store(lds, b + 6, wt_xi(v0, v1, 0));
    wt_store(lds, b + 7, wt_xi(v1, v0, 1));
  }
}

}  // namespace snarkv
