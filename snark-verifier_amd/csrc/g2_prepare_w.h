// Wavefront-parallel G2 line tables: the `G2Prepared::from` of the reference's decider (pcs/kzg/decider.rs:74, done here once
// per deciding key), i.e. pairing.h g2_prepare -- 64 doubling + 38 addition steps on the twist, projective, inversion-free.
// Round 1-3 ran it on ONE lane in the 8 x 32-bit field (~3 000 dependent Fq products of ~600 instructions: 5.1 ms per key,
// 1.4 KB of stack).  Here the formulas are cut into LEVELS of mutually independent Fq2 products (g2_prepare_prog.inc, made by
// gen_g2_prepare_prog.py): a lane computes ONE component of one product as a fused two-product Montgomery step on the lazy
// 29-bit field, its operands small integer combinations of LDS slots, so additions and subtractions cost no level.  Both
// points of a key (g2, -s_g2) run side by side in one wavefront, 14 lanes each: 448 levels of ~600 instructions.
// The output is the 29-bit table the decide kernels read (G2Prepared29), bit for bit what k_g2_to29(g2_prepare()) gave.
// Host-compilable: tests/hosttest emulates the lanes against pairing.h.
#pragma once
#include "decide_w.h"

namespace snarkv {

#include "g2_prepare_prog.inc"

struct Fq2_29P {  // one LDS slot: an Fq2 value, components carry-normalised
  Fq29 c[2];
};

// sum_k coeff_k * slot_k (component e), carry-normalised; coefficients in -3 .. 3, values within ~1.5 p: the sum within ~9 p
SNARKV_HD Fq29 g2w_comb(const Fq2_29P* sl, const int8_t s[3], const int8_t c[3], int e) {
  Fq29 r = fq29_zero();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int32_t ck = c[k];
    if (ck != 0) {
      const Fq29& v = sl[s[k]].c[e];
#pragma unroll
      for (int i = 0; i < 9; ++i) r.v[i] += ck * v.v[i];
    }
  }
  return fq29_norm(r);
}

// component e of the task's product (before the store / the canonical output)
SNARKV_HD Fq29 g2w_task(const Fq2_29P* sl, const G2wTask& t, int e) {
  Fq29 a0 = g2w_comb(sl, t.as, t.ac, 0), a1 = g2w_comb(sl, t.as, t.ac, 1);
  if (t.conj) a1 = fq29_neg(a1);
  const Fq29 b0 = g2w_comb(sl, t.bs, t.bc, 0), b1 = g2w_comb(sl, t.bs, t.bc, 1);
  // (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u
  return e ? fq29_mul2(a0, b1, a1, b0) : fq29_mul2(a0, b0, fq29_neg(a1), b1);
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
