// TEST INFRASTRUCTURE: compiles the device math headers (csrc/*.h) for the
// HOST with g++ so tests/ can unit-test the exact device functions without a
// GPU.  Never linked into the product library.
#include <cstring>
#include "../../snark-verifier_amd/csrc/g1.h"
#include "../../snark-verifier_amd/csrc/pairing.h"
#include "../../snark-verifier_amd/csrc/g1_29.h"
#include "../../snark-verifier_amd/csrc/glv.h"
#include "../../snark-verifier_amd/csrc/pairing_coop.h"
#include "../../snark-verifier_amd/csrc/pairing_coop29.h"
#include "../../snark-verifier_amd/csrc/fr29.h"
#include "../../snark-verifier_amd/csrc/decide_sched.hpp"
#include "../../snark-verifier_amd/csrc/g2_prepare_w.h"
#include <vector>

using namespace snarkv;

static Fq load_fq(const uint8_t* b) {
  uint32_t w[8];
  memcpy(w, b, 32);
  return fq_from_canonical(w);
}
static void store_fq(const Fq& a, uint8_t* b) {
  uint32_t w[8];
  fq_to_canonical(a, w);
  memcpy(b, w, 32);
}
static Fq2 load_fq2(const uint8_t* b) { return Fq2{load_fq(b), load_fq(b + 32)}; }
static void store_fq2(const Fq2& a, uint8_t* b) {
  store_fq(a.c0, b);
  store_fq(a.c1, b + 32);
}
static G1Affine load_g1(const uint8_t* b) {
  uint32_t w[16];
  memcpy(w, b, 64);
  return g1a_from_canonical(w);
}
static void store_g1(const G1Affine& p, uint8_t* b) {
  uint32_t w[16];
  g1a_to_canonical(p, w);
  memcpy(b, w, 64);
}
static void store_fq12(const Fq12& f, uint8_t* b) {
  const Fq6* h[2] = {&f.c0, &f.c1};
  for (int i = 0; i < 2; ++i) {
    store_fq2(h[i]->c0, b + (i * 3 + 0) * 64);
    store_fq2(h[i]->c1, b + (i * 3 + 1) * 64);
    store_fq2(h[i]->c2, b + (i * 3 + 2) * 64);
  }
}
static Fq12 load_fq12(const uint8_t* b) {
  Fq12 f;
  Fq6* h[2] = {&f.c0, &f.c1};
  for (int i = 0; i < 2; ++i) {
    h[i]->c0 = load_fq2(b + (i * 3 + 0) * 64);
    h[i]->c1 = load_fq2(b + (i * 3 + 1) * 64);
    h[i]->c2 = load_fq2(b + (i * 3 + 2) * 64);
  }
  return f;
}

extern "C" {
void ht_fq_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) { store_fq(fq_mul(load_fq(a), load_fq(b)), out); }
void ht_fq_add(const uint8_t* a, const uint8_t* b, uint8_t* out) { store_fq(fq_add(load_fq(a), load_fq(b)), out); }
void ht_fq_sub(const uint8_t* a, const uint8_t* b, uint8_t* out) { store_fq(fq_sub(load_fq(a), load_fq(b)), out); }
void ht_fq_inv(const uint8_t* a, uint8_t* out) { store_fq(fq_inv(load_fq(a)), out); }
int ht_g1_on_curve(const uint8_t* p) { return g1a_is_on_curve(load_g1(p)) ? 1 : 0; }
void ht_g1_add(const uint8_t* p, const uint8_t* q, uint8_t* out) {
  G1Xyzz a = xyzz_from_affine(load_g1(p));
  xyzz_add_mixed(a, load_g1(q));
  store_g1(xyzz_to_affine(a), out);
}
// (k1*P) + (k2*Q) through the full projective adder
void ht_g1_lincomb(const uint8_t* p, const uint8_t* k1, const uint8_t* q, const uint8_t* k2, uint8_t* out) {
  uint32_t w1[8], w2[8];
  memcpy(w1, k1, 32);
  memcpy(w2, k2, 32);
  G1Xyzz a = g1_scalar_mul(load_g1(p), w1);
  G1Xyzz b = g1_scalar_mul(load_g1(q), w2);
  xyzz_add(a, b);
  store_g1(xyzz_to_affine(a), out);
}
void ht_g1_mul(const uint8_t* p, const uint8_t* k, uint8_t* out) {
  uint32_t w[8];
  memcpy(w, k, 32);
  store_g1(xyzz_to_affine(g1_scalar_mul(load_g1(p), w)), out);
}
void ht_fq12_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  store_fq12(fq12_mul(load_fq12(a), load_fq12(b)), out);
}
void ht_fq12_sqr(const uint8_t* a, uint8_t* out) { store_fq12(fq12_sqr(load_fq12(a)), out); }
void ht_fq12_inv(const uint8_t* a, uint8_t* out) { store_fq12(fq12_inv(load_fq12(a)), out); }
void ht_fq12_frob(const uint8_t* a, int k, uint8_t* out) { store_fq12(fq12_frobenius(load_fq12(a), k), out); }
void ht_final_exp(const uint8_t* a, uint8_t* out) { store_fq12(final_exponentiation(load_fq12(a)), out); }
// e(P1,Q1)*e(P2,Q2) fully exponentiated; npairs in {1,2}
void ht_pairing_product(const uint8_t* p, const uint8_t* q, int npairs, uint8_t* out) {
  G2Prepared* prep = new G2Prepared[npairs];
  G1AffineM ps[2];
  const G2Prepared* qs[2];
  for (int k = 0; k < npairs; ++k) {
    G2Affine qa{load_fq2(q + 128 * k), load_fq2(q + 128 * k + 64)};
    g2_prepare(qa, prep[k]);
    G1Affine pa = load_g1(p + 64 * k);
    ps[k] = G1AffineM{pa.x, pa.y};
    qs[k] = &prep[k];
  }
  Fq12 f = multi_miller_loop(ps, qs, npairs);
  store_fq12(final_exponentiation(f), out);
  delete[] prep;
}

// ---------------- 9x29-bit lazy field / group (fq29.h, g1_29.h) ----------
static Fq29 load29(const uint8_t* b) {
  uint32_t w[8];
  memcpy(w, b, 32);
  return fq29_from_canonical(w);
}
static void store29(const Fq29& a, uint8_t* b) {
  uint32_t w[8];
  fq29_to_canonical(a, w);
  memcpy(b, w, 32);
}
static G1Affine29 load_g1_29(const uint8_t* b) {
  uint32_t w[16];
  memcpy(w, b, 64);
  return g1a29_from_canonical(w);
}
static void store_g1_29(const G1Affine29& p, uint8_t* b) {
  uint32_t w[16];
  g1a29_to_canonical(p, w);
  memcpy(b, w, 64);
}
void ht29_fq_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) { store29(fq29_mul(load29(a), load29(b)), out); }
void ht29_fq_sqr(const uint8_t* a, uint8_t* out) { store29(fq29_sqr(fq29_norm(load29(a))), out); }
void ht29_fq_inv(const uint8_t* a, uint8_t* out) { store29(fq29_inv(fq29_norm(load29(a))), out); }
// plain-integer inverses (8 x u32 LE in, out): safegcd (shipped) and the binary Euclid cross-check
void ht_words_inv_safegcd(const uint8_t* a, uint8_t* out) {
  uint32_t w[8], r[8];
  memcpy(w, a, 32);
  fq_words_inv_safegcd(w, r);
  memcpy(out, r, 32);
}
void ht_words_inv_binary(const uint8_t* a, uint8_t* out) {
  uint32_t w[8], r[8];
  memcpy(w, a, 32);
  fq_words_inv_binary(w, r);
  memcpy(out, r, 32);
}
void ht29_fq_inv_fermat(const uint8_t* a, uint8_t* out) { store29(fq29_inv_fermat(fq29_norm(load29(a))), out); }
// inverse of the lazy (unnormalised, possibly negative) difference a - b
void ht29_fq_inv_of_diff(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  store29(fq29_inv(fq29_sub(load29(a), load29(b))), out);
}
// ((a - b) * (c + d) - e) with lazy add/sub feeding products
void ht29_fq_lazy_expr(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, const uint8_t* e,
                       uint8_t* out) {
  Fq29 t = fq29_sub(load29(a), load29(b));
  Fq29 u = fq29_norm(fq29_add(load29(c), load29(d)));
  store29(fq29_sub(fq29_mul(t, u), load29(e)), out);
}
int ht29_is_zero_mod_p_of_diff(const uint8_t* a, const uint8_t* b) {
  return fq29_is_zero_mod_p(fq29_sub(load29(a), load29(b))) ? 1 : 0;
}
// sum of n affine points (sign[i] != 0 negates) by a madd chain; mode 0 = fast,
// 1 = careful.  Returns 1 if the fast result is degenerate.
int ht29_madd_chain(const uint8_t* pts, const uint8_t* sign, int n, int careful, uint8_t* out) {
  G1Xyzz29 acc = xyzz29_identity();
  bool started = false;
  for (int i = 0; i < n; ++i) {
    G1Affine29 p = load_g1_29(pts + 64 * i);
    if (sign[i]) p = g1a29_neg(p);
    if (careful) {
      xyzz29_madd_careful(acc, p);
    } else if (!started) {
      acc = xyzz29_from_affine(p);
      started = true;
    } else {
      xyzz29_madd_fast(acc, p);
    }
  }
  // a started fast run can never legitimately hold the identity: ZZ = 0 (mod p),
  // exact zero limbs included, means "redo carefully"
  int deg = (!careful && started && xyzz29_is_degenerate(acc)) ? 1 : 0;
  store_g1_29(xyzz29_to_affine(acc), out);
  return deg;
}
// (sum of first h points) + (sum of the rest) through the full adder
int ht29_add_halves(const uint8_t* pts, int n, int h, int careful, uint8_t* out) {
  G1Xyzz29 a = xyzz29_identity(), b = xyzz29_identity();
  for (int i = 0; i < n; ++i) xyzz29_madd_careful(i < h ? a : b, load_g1_29(pts + 64 * i));
  bool bad = false;
  if (careful) xyzz29_add_careful(a, b);
  else xyzz29_add_skipid_fast(a, b, bad);
  int deg = (!careful && (bad || (!xyzz29_is_identity(a) && xyzz29_is_degenerate(a)))) ? 1 : 0;
  store_g1_29(xyzz29_to_affine(a), out);
  return deg;
}
void ht29_g1_mul(const uint8_t* p, const uint8_t* k, int careful, uint8_t* out) {
  uint32_t w[8];
  memcpy(w, k, 32);
  G1Xyzz29 r = careful ? g1_29_scalar_mul<true>(load_g1_29(p), w) : g1_29_scalar_mul<false>(load_g1_29(p), w);
  store_g1_29(xyzz29_to_affine(r), out);
}
void ht_glv_decompose(const uint8_t* k, uint8_t* out32) {
  uint32_t w[8], o[8];
  memcpy(w, k, 32);
  glv_decompose(w, o);
  memcpy(out32, o, 32);
}
// phi(P) = (beta x, y) through the 29-bit field
void ht29_glv_phi(const uint8_t* p, uint8_t* out) {
  G1Affine29 a = load_g1_29(p);
  constexpr int32_t b[9] = SNARKV_GLV_BETA29_LIMBS;
  Fq29 beta;
  for (int i = 0; i < 9; ++i) beta.v[i] = b[i];
  a.x = fq29_mul(a.x, beta);
  store_g1_29(a, out);
}
// 2^n * P via the Jacobian repeated-doubling chain
void ht29_double_n(const uint8_t* p, int n, uint8_t* out) {
  G1Xyzz29 a = xyzz29_from_affine(load_g1_29(p));
  // start from a non-trivial XYZZ representation: a = P + P' - P' would need more plumbing; use 2P (ZZ != 1)
  a = xyzz29_double(a);
  store_g1_29(xyzz29_to_affine(xyzz29_double_n(a, n)), out);
}
// the round k_decide runs (pairing_coop29.h "coop3"): 96 lanes emulated one by
// one, butterflies replaced by explicit sums.  mode 0: f <- f*b each round;
// mode 1: f <- f*f, then f <- f*b (the Miller-loop pattern, both operands lazy).
static void coop3_round(const Fq29* fa, const Fq29* fb, Fq29* fc) {
  Fq29 prod[96];
  Coop3Lane L[96];
  for (int l = 0; l < 96; ++l) {
    L[l] = coop3_lane(l);
    prod[l] = L[l].active ? coop3_product(L[l].e, fa[2 * L[l].i1], fa[2 * L[l].i1 + 1], fb[2 * L[l].i2 + L[l].e],
                                          fb[2 * L[l].i2 + 1 - L[l].e])
                          : fq29_zero();
  }
  Fq29 lo[12], hi[12];
  for (int g = 0; g < 12; ++g) {
    lo[g] = fq29_zero();
    hi[g] = fq29_zero();
    for (int j = 0; j < 8; ++j) {
      int l = 8 * g + j;  // g = 2k + e
      if (!L[l].active) continue;
      Fq29& dst = L[l].high ? hi[g] : lo[g];
      for (int i = 0; i < 9; ++i) dst.v[i] = (int32_t)((uint32_t)dst.v[i] + (uint32_t)prod[l].v[i]);
    }
  }
  for (int g = 0; g < 12; ++g) fc[g] = coop3_finalize(g & 1, lo[g], hi[g], hi[g ^ 1]);
}
void ht_coop3_fq12_mul_iter(const uint8_t* a, const uint8_t* b, int rounds, int mode, uint8_t* out) {
  Fq fa8[12], fb8[12];
  coop_flat_from_tower(load_fq12(a), fa8);
  coop_flat_from_tower(load_fq12(b), fb8);
  Fq29 fa[12], fb[12], ft[12];
  for (int c = 0; c < 12; ++c) {
    uint32_t w[8];
    fq_to_canonical(fa8[c], w);
    fa[c] = fq29_canon_residue(fq29_from_canonical(w));
    fq_to_canonical(fb8[c], w);
    fb[c] = fq29_canon_residue(fq29_from_canonical(w));
  }
  for (int r = 0; r < rounds; ++r) {
    if (mode == 1) {
      coop3_round(fa, fa, ft);
      for (int c = 0; c < 12; ++c) fa[c] = ft[c];
    }
    coop3_round(fa, fb, ft);
    for (int c = 0; c < 12; ++c) fa[c] = ft[c];
  }
  Fq fc[12];
  for (int c = 0; c < 12; ++c) {
    uint32_t w[8];
    fq29_to_canonical(fa[c], w);
    fc[c] = fq_from_canonical(w);
  }
  store_fq12(coop_tower_from_flat(fc), out);
}
// ---------------- the program-driven decide kernel (decide_w.h, decide_sched.hpp), emulated lane by lane -----------------
// q2 = the two G2 points of the pairing product (256 bytes canonical; the SECOND one is used as given: pass -s_g2),
// acc = lhs || rhs.  Runs the set-up and every round of the scheduled program exactly as k_decide_w does -- 4 wavefronts of
// 64 lanes, tasks / squeeze / xi copies by the device functions, the DPP exchanges as explicit sums -- with every store of a
// round applied after all its loads (the scheduler's contract: no operation writes what its round reads).
// info[0..3] = rounds, LDS registers used, critical path, operations.  Returns 0, or -1 if a round violates the contract.
int ht_decide_w(const uint8_t* q2, const uint8_t* acc, uint8_t* out, int* info) {
  static WtProgram prog = wt_build_program();
  std::vector<G2Prepared29> prep(2);
  for (int k = 0; k < 2; ++k) {
    G2Prepared* p8 = new G2Prepared;
    G2Affine qa{load_fq2(q2 + 128 * k), load_fq2(q2 + 128 * k + 64)};
    g2_prepare(qa, *p8);
    prep[k].is_identity = p8->is_identity;
    for (int idx = 0; idx < kLinesPerG2 && !p8->is_identity; ++idx) {
      const LineCoeff& l = p8->line[idx];
      const Fq* src[6] = {&l.cy.c0, &l.cy.c1, &l.cx.c0, &l.cx.c1, &l.cw.c0, &l.cw.c1};
      for (int c = 0; c < 6; ++c) {
        uint32_t w[8];
        fq_to_canonical(*src[c], w);
        prep[k].line[idx].c[c] = fq29_canon_residue(fq29_from_canonical(w));
      }
    }
    delete p8;
  }
  std::vector<Fq29P> lds(kWtValues);
  memset(lds.data(), 0, lds.size() * sizeof(Fq29P));
  Fq29 pt[2][2];
  int live[2];
  for (int t = 0; t < 4; ++t) pt[t >> 1][t & 1] = fq29_canon_residue(load29(acc + 32 * t));
  for (int j = 0; j < 52; ++j) wt_store(lds.data(), kWtConstBase + j, wt_const_value(j));
  for (int k = 0; k < 2; ++k)
    live[k] = !(fq29_limbs_all_zero(pt[k][0]) && fq29_limbs_all_zero(pt[k][1])) && !prep[k].is_identity;
  for (int t = 0; t < 2 * kLinesPerG2 * 3; ++t) {
    const int pair = t / (kLinesPerG2 * 3), rem = t % (kLinesPerG2 * 3);
    wt_eval_line(lds.data(), prep.data(), pair, rem / 3, rem % 3, pt[pair][0], pt[pair][1], live[pair] != 0);
  }
  int nops = 0;
  for (int r = 0; r < prog.rounds; ++r) {
    std::vector<Fq29P> next = lds;  // stores land here
    for (int duo = 0; duo < 2; ++duo) {
      const WtOp op = prog.ops[2 * r + duo];
      if (op.kind == WT_IDLE) continue;
      ++nops;
      // contract: the destination is not read by either operation of this round
      for (int d2 = 0; d2 < 2; ++d2) {
        const WtOp o2 = prog.ops[2 * r + d2];
        if (o2.kind == WT_IDLE) continue;
        const int lo = op.dst, hi = op.dst + (op.kind == WT_FQ2INV ? 2 : kWtDense);
        if ((o2.a >= lo && o2.a < hi) || (o2.b >= lo && o2.b < hi)) return -1;
        if (d2 != duo && o2.dst == op.dst) return -1;
      }
      if (op.kind == WT_FQ2INV) {
        wt_fq2inv(next.data(), op);  // reads a's coefficient 0 (unchanged in `next`: nothing else wrote it), writes the scalar register
        continue;
      }
      for (int half = 0; half < 2; ++half) {
        Fq29 own[8];
        for (int g = 0; g < 8; ++g) {
          Fq29 s = fq29_zero();
          for (int jj = 0; jj < 8; ++jj) {
            const Fq29 v = wt_task(lds.data(), op, half, 8 * g + jj);
            for (int q = 0; q < 9; ++q) s.v[q] = (int32_t)((uint32_t)s.v[q] + (uint32_t)v.v[q]);
          }
          own[g] = wt_squeeze(s);
        }
        for (int g = 0; g < 8; ++g) wt_write(next.data(), op, half, 8 * g, own[g], own[g ^ 1]);
      }
    }
    lds.swap(next);
  }
  Fq fc[12];
  for (int c = 0; c < 12; ++c) {
    uint32_t w[8];
    fq29_to_canonical(wt_load(lds.data(), prog.result + c), w);
    fc[c] = fq_from_canonical(w);
  }
  store_fq12(coop_tower_from_flat(fc), out);
  if (info) info[0] = prog.rounds, info[1] = prog.regs_used, info[2] = prog.critical_path, info[3] = nops;
  return 0;
}
// ---------------- the wavefront-parallel G2 line tables (g2_prepare_w.h), emulated level by level ---------------------------
// q = one G2 point (128 bytes canonical).  Runs the generated level program exactly as k_g2_prepare_w does for one point
// (every task's two components from the slots as they were BEFORE the level, stores after) and compares every line
// coefficient with pairing.h g2_prepare brought to the same 29-bit canonical form.  Returns the number of mismatching
// coefficients (0 = identical tables), -1 for the identity.
int ht_g2_prepare_w(const uint8_t* q, int negate_y) {
  G2Affine qa{load_fq2(q), load_fq2(q + 64)};
  if (negate_y) qa.y = fq2_neg(qa.y);
  G2Prepared* ref = new G2Prepared;
  g2_prepare(qa, *ref);
  if (ref->is_identity) {
    delete ref;
    return -1;
  }
  std::vector<Fq2_29P> sl(kG2wSlots);
  for (int k = 0; k < kG2wSlots; ++k) g2w_put(sl.data(), k, 0, fq29_zero()), g2w_put(sl.data(), k, 1, fq29_zero());
  auto put = [&](int slot, const Fq2& v) {
    const Fq* c[2] = {&v.c0, &v.c1};
    for (int e = 0; e < 2; ++e) {
      uint32_t w[8];
      fq_to_canonical(*c[e], w);
      g2w_put(sl.data(), slot, e, fq29_canon_residue(fq29_from_canonical(w)));
    }
  };
  put(kG2wSlotQX, qa.x), put(kG2wSlotQY, qa.y), put(kG2wSlotTX, qa.x), put(kG2wSlotTYA, qa.y);
  for (int k = 0; k < 6; ++k)
    for (int e = 0; e < 2; ++e) g2w_put(sl.data(), k, e, g2w_const(k, e));
  g2w_put(sl.data(), kG2wSlotTZ, 0, fq29_one());
  std::vector<Fq29> lines(kLinesPerG2 * 6, fq29_zero());
  for (int lv = 0; lv < kG2wLevels; ++lv) {
    Fq29 val[kG2wTasks][2];
    for (int t = 0; t < kG2wTasks; ++t)
      for (int e = 0; e < 2; ++e)
        if (kG2wProg[lv][t].used) val[t][e] = g2w_task(sl.data(), kG2wProg[lv][t], e);
    for (int t = 0; t < kG2wTasks; ++t) {
      const G2wTask& tk = kG2wProg[lv][t];
      if (!tk.used) continue;
      for (int e = 0; e < 2; ++e) {
        if (tk.dst >= 0) g2w_put(sl.data(), tk.dst, e, val[t][e]);
        else lines[(tk.out / 3) * 6 + 2 * (tk.out % 3) + e] = fq29_canon_of_product(val[t][e]);
      }
    }
  }
  int bad = 0;
  for (int idx = 0; idx < kLinesPerG2; ++idx) {
    const LineCoeff& l = ref->line[idx];
    const Fq* src[6] = {&l.cy.c0, &l.cy.c1, &l.cx.c0, &l.cx.c1, &l.cw.c0, &l.cw.c1};
    for (int c = 0; c < 6; ++c) {
      uint32_t w[8];
      fq_to_canonical(*src[c], w);
      const Fq29 want = fq29_canon_residue(fq29_from_canonical(w));
      for (int i = 0; i < 9; ++i)
        if (want.v[i] != lines[idx * 6 + c].v[i]) {
          ++bad;
          break;
        }
    }
  }
  delete ref;
  return bad;
}

// ---------------- SNARKV_FLAG_MONTGOMERY codecs (fq29.h fq29_from_words / fq29_to_words, fr29.h fr_words_from_mont256) ---------
// in: 32 bytes; mode 0: canonical -> in-memory form (a * 2^256 mod p) through the 29-bit domain; 1: the reverse;
// 2: Fr in-memory -> canonical.  Exactly the device functions.
void ht_mont_codec(const uint8_t* in, int mode, uint8_t* out) {
  uint32_t w[8], o[8];
  memcpy(w, in, 32);
  if (mode == 0) fq29_to_words(fq29_from_words(w, false), o, true);
  else if (mode == 1) fq29_to_words(fq29_from_words(w, true), o, false);
  else fr_words_from_mont256(w, o);
  memcpy(out, o, 32);
}
// a point through g1a29_from_words / g1a29_to_words in the given encodings
void ht_g1_recode(const uint8_t* in64, int mont_in, int mont_out, uint8_t* out64) {
  uint32_t w[16], o[16];
  memcpy(w, in64, 64);
  g1a29_to_words(g1a29_from_words(w, mont_in != 0), o, mont_out != 0);
  memcpy(out64, o, 64);
}
void ht29_fq_mul2(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, int neg_c, uint8_t* out) {
  Fq29 cc = load29(c);
  if (neg_c) cc = fq29_neg(cc);
  store29(fq29_mul2(load29(a), load29(b), cc, load29(d)), out);
}
// scalar field on the 29-bit form (fr29.h): ((a * b) + c)^5, canonical in / out
void ht_fr29_expr(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* out) {
  uint32_t w[8];
  memcpy(w, a, 32);
  Fr29 x = fr29_from_canonical(w);
  memcpy(w, b, 32);
  Fr29 y = fr29_from_canonical(w);
  memcpy(w, c, 32);
  Fr29 z = fr29_from_canonical(w);
  Fr29 r = fr29_pow5(fr29_norm(fr29_add(fr29_mul(x, y), z)));
  fr29_to_canonical(r, w);
  memcpy(out, w, 32);
}
void ht_fr29_roundtrip(const uint8_t* a, uint8_t* out) {
  uint32_t w[8];
  memcpy(w, a, 32);
  Fr29 x = fr29_canon_residue(fr29_from_canonical(w));
  fr29_to_canonical(x, w);
  memcpy(out, w, 32);
}
}

// packed 64-byte memory form of the Pippenger's Montgomery points (g1_29.h G1Packed): canonical x | y -> Montgomery
// canonical residues -> pack -> unpack -> back to canonical; also reports the packed words and whether every unpacked
// limb is in [0, 2^29)
extern "C" int ht_g1_pack_roundtrip(const uint8_t* p64, uint8_t* out64, uint8_t* packed64) {
  uint32_t w[16];
  memcpy(w, p64, 64);
  G1Affine29 a = g1a29_from_canonical(w);
  G1Packed k = g1a29_pack(a);
  memcpy(packed64, k.w, 64);
  G1Affine29 b = g1a29_unpack(k);
  int ok = 1;
  for (int i = 0; i < 9; ++i) {
    ok &= b.x.v[i] == a.x.v[i] && b.y.v[i] == a.y.v[i];
    ok &= b.x.v[i] >= 0 && b.x.v[i] <= kMask29 && b.y.v[i] >= 0 && b.y.v[i] <= kMask29;
  }
  uint32_t o[16];
  g1a29_to_canonical(b, o);
  memcpy(out64, o, 64);
  return ok;
}

