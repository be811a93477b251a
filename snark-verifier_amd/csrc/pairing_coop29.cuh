// Lane-level pieces of the workgroup-cooperative pairing on the lazy 9x29-bit
// field (see pairing_coop.cuh for the round structure and the tables; this is
// the arithmetic the shipped `k_decide` uses).
//
// Why 29-bit here too: on the exact 8x32 field every signed sum of the two
// reduction stages is a dependent carry chain, and gfx950 needs two wait
// states between a VALU writing a carry and the VALU consuming it -- the
// cheap-looking additions cost as much as the products.  In the lazy form a
// sum is nine independent `v_add_u32`, and one float-estimated quotient
// (`fq29_reduce_small`) squeezes the 22-term coefficient back below 1.5p.
//
// Invariants of a coefficient c held in LDS: carry-normalised, |c| < 1.5p;
// c9 = 9c carry-normalised.  Products (a or 9a) * b then stay inside the mul
// budget (|limb| < 2^29 both sides) and every product value inside (-p/8, 9p/8).
#pragma once
#include "fq29.cuh"
#include "pairing_coop_tables.h"

namespace snarkv {

struct Fq2_29 {
  Fq29 c0, c1;
};

SNARKV_HD Fq2_29 frob29_gamma(int k, int i) {
  constexpr int32_t g1[5][2][9] = BN254_FROB29_GAMMA_1;
  constexpr int32_t g2[5][2][9] = BN254_FROB29_GAMMA_2;
  constexpr int32_t g3[5][2][9] = BN254_FROB29_GAMMA_3;
  Fq2_29 r;
  for (int l = 0; l < 9; ++l) {
    r.c0.v[l] = (k == 1) ? g1[i - 1][0][l] : (k == 2) ? g2[i - 1][0][l] : g3[i - 1][0][l];
    r.c1.v[l] = (k == 1) ? g1[i - 1][1][l] : (k == 2) ? g2[i - 1][1][l] : g3[i - 1][1][l];
  }
  return r;
}

// round 1, lane l < COOP_NPROD
SNARKV_HD Fq29 coop29_product(unsigned desc, const Fq29* a, const Fq29* a9, const Fq29* b) {
  unsigned s = desc & 15u, t = (desc >> 4) & 15u;
  return fq29_mul((desc >> 8) ? a9[s] : a[s], b[t]);
}

// round 2, lane q < 48: signed sum of <= 6 products, three at a time so the
// lazy limbs stay below 2^31; result carry-normalised
SNARKV_HD Fq29 coop29_stage1(const unsigned short* ent, const Fq29* prods) {
  Fq29 h[2];
  for (int half = 0; half < 2; ++half) {
    Fq29 acc = fq29_zero();
    for (int k = 3 * half; k < 3 * half + 3; ++k) {
      unsigned e = ent[k];
      if (e == 0xFFFFu) continue;
      const Fq29& p = prods[e & 0x7FFFu];
      acc = (e & 0x8000u) ? fq29_sub(acc, p) : fq29_add(acc, p);
    }
    h[half] = fq29_norm(acc);
  }
  return fq29_norm(fq29_add(h[0], h[1]));
}

// round 3, lane c < 12: sum of four partials, modular squeeze
SNARKV_HD Fq29 coop29_stage2(int c, const Fq29* parts) {
  Fq29 s = fq29_add(fq29_add(parts[4 * c], parts[4 * c + 1]), fq29_add(parts[4 * c + 2], parts[4 * c + 3]));
  return fq29_reduce_small(fq29_norm(s));
}

SNARKV_HD Fq29 coop29_times9(const Fq29& c) { return fq29_mul_small_norm(c, 9); }

}  // namespace snarkv
