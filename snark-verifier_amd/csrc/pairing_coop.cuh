// Lane-level pieces of the workgroup-cooperative pairing (decider_coop.hip).
//
// Why: a pairing is ~20 000 dependent Fq products.  One lane per pairing
// (decider.hip v1) leaves a single `decide` at 45-55 ms on MI355X -- 30x slower
// than a CPU core -- because a lone lane issues one instruction every ~5
// cycles.  Here ONE workgroup of 256 lanes serves ONE accumulator and every
// Fq12 multiplication is a single parallel round:
//   round 1  204 lanes: one Fq product each  (144 a_s b_t  + 60 (9 a_s) b_t for
//            the terms that wrap through w^6 = 9 + u)
//   round 2  48 lanes: signed sums of <= 6 products  (tables: gen_coop_tables.py)
//   round 3  12 lanes: sum of 4 partials -> coefficient c, and 9c for later
// Flat basis: coefficient index c = 2 i + e  <->  u^e w^i.
// These functions are host-compilable so tests/hosttest can emulate the lanes
// and validate the tables against the tower arithmetic.
#pragma once
#include "pairing.cuh"
#include "pairing_coop_tables.h"

namespace snarkv {

// tower (c0.{c0,c1,c2}, c1.{c0,c1,c2}) <-> flat: w^0..w^5 = c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2
SNARKV_HD void coop_flat_from_tower(const Fq12& f, Fq flat[12]) {
  const Fq2* g[6] = {&f.c0.c0, &f.c1.c0, &f.c0.c1, &f.c1.c1, &f.c0.c2, &f.c1.c2};
  for (int i = 0; i < 6; ++i) {
    flat[2 * i] = g[i]->c0;
    flat[2 * i + 1] = g[i]->c1;
  }
}
SNARKV_HD Fq12 coop_tower_from_flat(const Fq flat[12]) {
  Fq12 f;
  Fq2* g[6] = {&f.c0.c0, &f.c1.c0, &f.c0.c1, &f.c1.c1, &f.c0.c2, &f.c1.c2};
  for (int i = 0; i < 6; ++i) {
    g[i]->c0 = flat[2 * i];
    g[i]->c1 = flat[2 * i + 1];
  }
  return f;
}

SNARKV_HD Fq fq_mul9(const Fq& x) {
  Fq x2 = fq_dbl(x), x4 = fq_dbl(x2), x8 = fq_dbl(x4);
  return fq_add(x8, x);
}

// round 1, lane l < COOP_NPROD
SNARKV_HD Fq coop_product(int l, const Fq* a, const Fq* a9, const Fq* b) {
  unsigned e = kCoopProd[l];
  unsigned s = e & 15u, t = (e >> 4) & 15u;
  return fq_mul((e >> 8) ? a9[s] : a[s], b[t]);
}

// round 2, lane q < 48: partial sum for coefficient q/4
SNARKV_HD Fq coop_stage1(int q, const Fq* prods) {
  Fq acc = fq_zero();
  for (int k = 0; k < COOP_STAGE1_TERMS; ++k) {
    unsigned e = kCoopStage1[q][k];
    if (e == 0xFFFFu) break;
    const Fq& p = prods[e & 0x7FFFu];
    acc = (e & 0x8000u) ? fq_sub(acc, p) : fq_add(acc, p);
  }
  return acc;
}

// round 3, lane c < 12
SNARKV_HD Fq coop_stage2(int c, const Fq* parts) {
  return fq_add(fq_add(parts[4 * c], parts[4 * c + 1]), fq_add(parts[4 * c + 2], parts[4 * c + 3]));
}

}  // namespace snarkv
