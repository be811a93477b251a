// BN254 G1 (y^2 = x^3 + 3 over Fq) group law for the device.
//
// Replaces, on the device, the halo2curves `G1Affine`/`G1` operations the
// reference's hot path calls: `*base * scalar`, `acc + value`, `to_affine`
// (reference `snark-verifier/src/loader/native.rs:67-70`) and `lhs + rhs`,
// `+=`, `double`, `identity` (reference `snark-verifier/src/util/msm.rs:239-254,
// 286,298-301`).  halo2curves works in Jacobian coordinates; the boundary only
// ever sees canonical affine bytes, so the
// projective system is free.  We use extended Jacobian "XYZZ"
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): mixed add 8M+2S, add 12M+2S, no
// inversion; identity <=> ZZ == 0.
#pragma once
#include "fq.h"

namespace snarkv {

struct G1Affine {  // identity = (0, 0), as halo2curves
  Fq x, y;
};

struct G1Xyzz {
  Fq x, y, zz, zzz;
};

SNARKV_HD bool g1a_is_identity(const G1Affine& p) { return fq_is_zero(p.x) && fq_is_zero(p.y); }

SNARKV_HD G1Xyzz xyzz_identity() {
  G1Xyzz r;
  r.x = fq_zero();
  r.y = fq_zero();
  r.zz = fq_zero();
  r.zzz = fq_zero();
  return r;
}

SNARKV_HD bool xyzz_is_identity(const G1Xyzz& p) { return fq_is_zero(p.zz); }

SNARKV_HD G1Xyzz xyzz_from_affine(const G1Affine& p) {
  G1Xyzz r;
  if (g1a_is_identity(p)) return xyzz_identity();
  r.x = p.x;
  r.y = p.y;
  r.zz = fq_one();
  r.zzz = fq_one();
  return r;
}

SNARKV_HD G1Affine g1a_neg(const G1Affine& p) {
  G1Affine r;
  r.x = p.x;
  r.y = fq_neg(p.y);  // -(0) = 0 keeps the identity
  return r;
}

// 2*P for affine P (mdbl-2008-s-1, a = 0).
SNARKV_HD G1Xyzz xyzz_double_affine(const G1Affine& p) {
  if (g1a_is_identity(p)) return xyzz_identity();
  G1Xyzz r;
  Fq u = fq_dbl(p.y);
  Fq v = fq_sqr(u);
  Fq w = fq_mul(u, v);
  Fq s = fq_mul(p.x, v);
  Fq x2 = fq_sqr(p.x);
  Fq m = fq_add(fq_dbl(x2), x2);
  r.x = fq_sub(fq_sqr(m), fq_dbl(s));
  r.y = fq_sub(fq_mul(m, fq_sub(s, r.x)), fq_mul(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// 2*P (dbl-2008-s-1, a = 0).  Identity (ZZ = 0) stays identity.
SNARKV_HD G1Xyzz xyzz_double(const G1Xyzz& p) {
  G1Xyzz r;
  Fq u = fq_dbl(p.y);
  Fq v = fq_sqr(u);
  Fq w = fq_mul(u, v);
  Fq s = fq_mul(p.x, v);
  Fq x2 = fq_sqr(p.x);
  Fq m = fq_add(fq_dbl(x2), x2);
  r.x = fq_sub(fq_sqr(m), fq_dbl(s));
  r.y = fq_sub(fq_mul(m, fq_sub(s, r.x)), fq_mul(w, p.y));
  r.zz = fq_mul(v, p.zz);
  r.zzz = fq_mul(w, p.zzz);
  return r;
}

// acc += P, P affine (madd-2008-s) with the exceptional cases made explicit:
// P = O, acc = O, acc = P (doubling), acc = -P (-> O).  Duplicate and
// opposite bases are legal inputs (SURVEY.md section 7 "Exceptional cases").
SNARKV_HD void xyzz_add_mixed(G1Xyzz& acc, const G1Affine& p) {
  if (g1a_is_identity(p)) return;
  if (xyzz_is_identity(acc)) {
    acc.x = p.x;
    acc.y = p.y;
    acc.zz = fq_one();
    acc.zzz = fq_one();
    return;
  }
  Fq u2 = fq_mul(p.x, acc.zz);
  Fq s2 = fq_mul(p.y, acc.zzz);
  Fq pp_ = fq_sub(u2, acc.x);
  Fq r = fq_sub(s2, acc.y);
  if (fq_is_zero(pp_)) {
    if (fq_is_zero(r)) {
      acc = xyzz_double_affine(p);
    } else {
      acc = xyzz_identity();
    }
    return;
  }
  Fq pp = fq_sqr(pp_);
  Fq ppp = fq_mul(pp_, pp);
  Fq q = fq_mul(acc.x, pp);
  Fq x3 = fq_sub(fq_sub(fq_sqr(r), ppp), fq_dbl(q));
  Fq y3 = fq_sub(fq_mul(r, fq_sub(q, x3)), fq_mul(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = fq_mul(acc.zz, pp);
  acc.zzz = fq_mul(acc.zzz, ppp);
}

// acc += b (add-2008-s) with the same exceptional cases.
SNARKV_HD void xyzz_add(G1Xyzz& acc, const G1Xyzz& b) {
  if (xyzz_is_identity(b)) return;
  if (xyzz_is_identity(acc)) {
    acc = b;
    return;
  }
  Fq u1 = fq_mul(acc.x, b.zz);
  Fq u2 = fq_mul(b.x, acc.zz);
  Fq s1 = fq_mul(acc.y, b.zzz);
  Fq s2 = fq_mul(b.y, acc.zzz);
  Fq pp_ = fq_sub(u2, u1);
  Fq r = fq_sub(s2, s1);
  if (fq_is_zero(pp_)) {
    if (fq_is_zero(r)) {
      acc = xyzz_double(acc);
    } else {
      acc = xyzz_identity();
    }
    return;
  }
  Fq pp = fq_sqr(pp_);
  Fq ppp = fq_mul(pp_, pp);
  Fq q = fq_mul(u1, pp);
  Fq x3 = fq_sub(fq_sub(fq_sqr(r), ppp), fq_dbl(q));
  Fq y3 = fq_sub(fq_mul(r, fq_sub(q, x3)), fq_mul(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = fq_mul(fq_mul(acc.zz, b.zz), pp);
  acc.zzz = fq_mul(fq_mul(acc.zzz, b.zzz), ppp);
}

// `to_affine()`: x = X/ZZ, y = Y/ZZZ with one inversion of ZZ*ZZZ.
SNARKV_HD G1Affine xyzz_to_affine(const G1Xyzz& p) {
  G1Affine r;
  if (xyzz_is_identity(p)) {
    r.x = fq_zero();
    r.y = fq_zero();
    return r;
  }
  Fq i = fq_inv(fq_mul(p.zz, p.zzz));
  Fq izz = fq_mul(i, p.zzz);  // 1/ZZ
  Fq izzz = fq_mul(i, p.zz);  // 1/ZZZ
  r.x = fq_mul(p.x, izz);
  r.y = fq_mul(p.y, izzz);
  return r;
}

// y^2 == x^3 + 3 (Montgomery domain); the identity (0,0) is accepted.
SNARKV_HD bool g1a_is_on_curve(const G1Affine& p) {
  if (g1a_is_identity(p)) return true;
  constexpr uint32_t three[8] = SNARKV_G1_B_MONT;
  Fq b;
#pragma unroll
  for (int i = 0; i < 8; ++i) b.v[i] = three[i];
  Fq lhs = fq_sqr(p.y);
  Fq rhs = fq_add(fq_mul(fq_sqr(p.x), p.x), b);
  return fq_eq(lhs, rhs);
}

// Boundary codec: 64-byte x||y canonical little-endian <-> Montgomery affine.
SNARKV_HD G1Affine g1a_from_canonical(const uint32_t w[16]) {
  G1Affine r;
  r.x = fq_from_canonical(w);
  r.y = fq_from_canonical(w + 8);
  return r;
}

SNARKV_HD void g1a_to_canonical(const G1Affine& p, uint32_t w[16]) {
  fq_to_canonical(p.x, w);
  fq_to_canonical(p.y, w + 8);
}

// k*P by left-to-right double-and-add over the 256-bit canonical scalar
// (8 LE words) -- the per-term operation of
// `NativeLoader::multi_scalar_multiplication` (reference native.rs:67).
SNARKV_HD G1Xyzz g1_scalar_mul(const G1Affine& p, const uint32_t k[8]) {
  G1Xyzz acc = xyzz_identity();
  for (int i = 7; i >= 0; --i) {
    uint32_t w = k[i];
    for (int b = 31; b >= 0; --b) {
      acc = xyzz_double(acc);
      if ((w >> b) & 1u) xyzz_add_mixed(acc, p);
    }
  }
  return acc;
}

}  // namespace snarkv
