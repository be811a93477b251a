// Flat basis of the workgroup-cooperative pairing: coefficient index
// c = 2 i + e  <->  u^e w^i  (the rounds themselves: pairing_coop29.h, decider.hip).
// Host-compilable so tests/hosttest can emulate the lanes against the tower arithmetic.
#pragma once
#include "pairing.h"

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

}  // namespace snarkv
