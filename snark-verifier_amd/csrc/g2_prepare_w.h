// Wavefront-parallel G2 line tables: the `G2Prepared::from` of the reference's decider (pcs/kzg/decider.rs:74, done here once
// per deciding key), i.e. pairing.h g2_prepare -- 64 doubling + 38 addition steps on the twist, projective, inversion-free.
// Round 1-3 ran it on ONE lane in the 8 x 32-bit field (~3 000 dependent Fq products of ~600 instructions: 5.1 ms per key,
// 1.4 KB of stack).  Here the formulas are cut into LEVELS of mutually independent Fq2 products (g2_prepare_prog.inc, made by
// gen_g2_prepare_prog.py): a lane computes ONE component of one product as a fused two-product Montgomery step on the lazy
// 29-bit field, its operands small integer combinations of LDS slots, so additions and subtractions cost no level.  Both
// points of a key (g2, -s_g2) run side by side in one wavefront, 14 lanes each: 448 levels of ~600 instructions.
// The output is the 29-bit table the decide kernels read (G2Prepared29), bit for bit what the one-lane g2_prepare() of rounds 1-3 gave, converted to 29-bit limbs.
// Host-compilable: tests/hosttest emulates the lanes against pairing.h.
#pragma once
#include "decide_w.h"

namespace snarkv {

#include "g2_prepare_prog.inc"

struct Fq2_29P {  // one LDS slot: an Fq2 value, components carry-normalised (8-byte aligned: 64-bit LDS reads)
  Fq29P c[2];
};
SNARKV_HD Fq29 g2w_get(const Fq2_29P* sl, int slot, int e) { return wt_load(sl[slot].c, e); }
SNARKV_HD void g2w_put(Fq2_29P* sl, int slot, int e, const Fq29& x) { wt_store(sl[slot].c, e, x); }

// sgn * sum_k coeff_k * slot_k (component e), carry-normalised; coefficients in -3 .. 3, values within ~1.5 p: the sum within ~9 p
SNARKV_HD Fq29 g2w_comb(const Fq2_29P* sl, const int8_t s[3], const int8_t c[3], int e, int32_t sgn) {
  Fq29 r = fq29_zero();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int32_t ck = c[k] * sgn;
    if (ck != 0) {
      const Fq29 v = g2w_get(sl, s[k], e);
#pragma unroll
      for (int i = 0; i < 9; ++i) r.v[i] += ck * v.v[i];
    }
  }
  return fq29_norm(r);
}

// A lane (task, e) builds component e of BOTH operands (the conjugate of A folded into its coefficients' sign); its
// neighbour builds the other component, and the two swap (one DPP move per limb on the device) instead of each building
// all four.
SNARKV_HD void g2w_mine(const Fq2_29P* sl, const G2wTask& t, int e, Fq29& am, Fq29& bm) {
  am = g2w_comb(sl, t.as, t.ac, e, (t.conj && e) ? -1 : 1);
  bm = g2w_comb(sl, t.bs, t.bc, e, 1);
}
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u, component e from my (am, bm) and my neighbour's (ap, bp)
// components: e = 0: am bm + (-ap) bp;  e = 1: ap bm + am bp -- ONE fused two-product step, its operands selected per lane
// (both forms behind a branch on e cost the wavefront two steps).
SNARKV_HD Fq29 g2w_product(const Fq29& am, const Fq29& bm, const Fq29& ap, const Fq29& bp, int e) {
  Fq29 x1, x2;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    x1.v[i] = e ? ap.v[i] : am.v[i];
    x2.v[i] = e ? am.v[i] : -ap.v[i];
  }
  return fq29_mul2(x1, bm, x2, bp);
}

// component e of the task's product, all on one lane (host emulation; the kernel swaps with the neighbour instead)
SNARKV_HD Fq29 g2w_task(const Fq2_29P* sl, const G2wTask& t, int e) {
  Fq29 am, bm, ap, bp;
  g2w_mine(sl, t, e, am, bm);
  g2w_mine(sl, t, e ^ 1, ap, bp);
  return g2w_product(am, bm, ap, bp, e);
}

// the constants of the program as 29-bit Montgomery residues (from the 8 x 32 Montgomery tables of bn254_consts.h)
SNARKV_HD Fq29 g2w_from_mont32(const uint32_t (&m)[8]) {
  Fq f;
  for (int i = 0; i < 8; ++i) f.v[i] = m[i];
  uint32_t w[8];
  fq_to_canonical(f, w);
  return fq29_canon_residue(fq29_from_canonical(w));
}
SNARKV_HD Fq29 g2w_const(int slot, int e) {
  constexpr uint32_t b3[2][8] = {BN254_TWIST_3B_C0_MONT, BN254_TWIST_3B_C1_MONT};
  constexpr uint32_t g12[2][8] = BN254_TWIST_G12;
  constexpr uint32_t g13[2][8] = BN254_TWIST_G13;
  constexpr uint32_t g22[2][8] = BN254_TWIST_G22;
  constexpr uint32_t g23[2][8] = BN254_TWIST_G23;
  if (slot == kG2wSlotONE) return e ? fq29_zero() : fq29_one();
  if (slot == kG2wSlotB3) return g2w_from_mont32(b3[e]);
  if (slot == kG2wSlotG12) return g2w_from_mont32(g12[e]);
  if (slot == kG2wSlotG13) return g2w_from_mont32(g13[e]);
  if (slot == kG2wSlotG22) return g2w_from_mont32(g22[e]);
  return g2w_from_mont32(g23[e]);
}

}  // namespace snarkv
