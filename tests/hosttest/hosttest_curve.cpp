// Host build of the CURVE-GENERIC device headers (fq29.h, fr29.h, g1_29.h, glv.h), compiled twice
// by tests/conftest.py: default (BN254) and -DSNARKV_CURVE_PALLAS.  Test infrastructure only.
#include <string.h>
#include "../../snark-verifier_amd/csrc/fr29.h"
#include "../../snark-verifier_amd/csrc/g1_29.h"
#include "../../snark-verifier_amd/csrc/glv.h"

using namespace snarkv;

static Fq29 ld(const uint8_t* b) {
  uint32_t w[8];
  memcpy(w, b, 32);
  return fq29_from_canonical(w);
}
static void st(const Fq29& a, uint8_t* b) {
  uint32_t w[8];
  fq29_to_canonical(a, w);
  memcpy(b, w, 32);
}
static G1Affine29 ldp(const uint8_t* b) {
  uint32_t w[16];
  memcpy(w, b, 64);
  return g1a29_from_canonical(w);
}
static void stp(const G1Xyzz29& p, uint8_t* out) {
  uint32_t w[16];
  g1a29_to_canonical(xyzz29_to_affine(p), w);
  memcpy(out, w, 64);
}

extern "C" {
const char* hc_curve() { return SNARKV_CURVE_NAME; }
void hc_fq_mul(const uint8_t* a, const uint8_t* b, uint8_t* o) { st(fq29_mul(ld(a), ld(b)), o); }
void hc_fq_sqr(const uint8_t* a, uint8_t* o) { st(fq29_sqr(ld(a)), o); }
void hc_fq_mul2(const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, uint8_t* o) {
  st(fq29_mul2(ld(a), ld(b), fq29_neg(ld(c)), ld(d)), o);  // a b - c d
}
void hc_fq_inv(const uint8_t* a, uint8_t* o) { st(fq29_inv(ld(a)), o); }
void hc_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* o) {
  uint32_t x[8], y[8], w[8];
  memcpy(x, a, 32);
  memcpy(y, b, 32);
  fr29_to_canonical(fr29_mul(fr29_from_canonical(x), fr29_from_canonical(y)), w);
  memcpy(o, w, 32);
}
// (P + Q) careful, P + Q fast (distinct, non-special inputs), 2^n P through the Jacobian chain
void hc_g1_add(const uint8_t* p, const uint8_t* q, uint8_t* o) {
  G1Xyzz29 a = xyzz29_from_affine(ldp(p));
  xyzz29_add_careful(a, xyzz29_from_affine(ldp(q)));
  stp(a, o);
}
void hc_g1_madd_chain(const uint8_t* p, const uint8_t* qs, int n, uint8_t* o) {
  G1Xyzz29 a = xyzz29_from_affine(ldp(p));
  for (int i = 0; i < n; ++i) xyzz29_madd_fast(a, ldp(qs + 64 * i));
  stp(a, o);
}
void hc_g1_double_n(const uint8_t* p, int n, uint8_t* o) {
  G1Xyzz29 a = xyzz29_double(xyzz29_from_affine(ldp(p)));  // a non-trivial ZZ to start from
  stp(xyzz29_double_n(a, n), o);
}
void hc_glv_decompose(const uint8_t* k, uint8_t* out32) {
  uint32_t w[8], o[8];
  memcpy(w, k, 32);
  glv_decompose(w, o);
  memcpy(out32, o, 32);
}
void hc_glv_phi(const uint8_t* p, uint8_t* out) {
  G1Affine29 a = ldp(p);
  constexpr int32_t b[9] = SNARKV_GLV_BETA29_LIMBS;
  Fq29 beta;
  for (int i = 0; i < 9; ++i) beta.v[i] = b[i];
  a.x = fq29_mul(a.x, beta);
  stp(xyzz29_from_affine(a), out);
}
}
