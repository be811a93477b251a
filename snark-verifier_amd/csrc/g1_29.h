// G1 group law over the lazy 9x29-bit field (fq29.h) -- the MSM hot loops.
//
// Two flavours of every adder:
//   *_fast     branch-free formulas, NO exceptional-case tests.  If an addition
//              ever meets P = +-Q (or a result is the identity) its ZZ becomes
//              = 0 (mod p) and stays so through every later addition (ZZ3 =
//              ZZ1*ZZ2*PP), so a caller checks `xyzz29_is_degenerate` ONCE at
//              the end of its run and, if set, redoes the run with
//   *_careful  the same formulas with the explicit P = Q / P = -Q / identity
//              cases (duplicate and opposite bases are legal inputs: SURVEY.md
//              section 7 "Exceptional cases").
// Stored identity = all limbs of ZZ zero (exact), tested with a cheap OR.
//
// Limb/value bounds kept by the formulas (N = product output: limbs 0..7 in
// [0,2^29), value in (-p/4, 5p/4)):
//   acc.x   carry-normalised, value in (-4p, 2p)
//   acc.y   lazy difference of two N: |limb| < 2^29, value in (-2p, 2p)
//   acc.zz, acc.zzz   N
// Every product below has one carry-normalised operand (< 2^29) and one with
// |limb| < 2^30, inside the 2^59.6 budget of fq29_mul.
#pragma once
#include "fq29.h"

namespace snarkv {

struct G1Affine29 {  // Montgomery (R = 2^261), canonical limbs; identity = all zero
  Fq29 x, y;
};

struct G1Xyzz29 {
  Fq29 x, y, zz, zzz;
};

// Memory form of the Montgomery points the Pippenger gathers: each coordinate's canonical residue (< p < 2^256) in
// 8 words (codecs below), 64 bytes per point, 64-byte aligned -- ONE 64-byte sector per gather where the
// 72-byte limb form straddled two (and, 7 times out of 8, two 128-byte lines): rocprofv3 FETCH_SIZE of k_accumulate
// halves.  The 9 x 29-bit limbs are cut out of the words in registers (a v_alignbit + v_and per limb).
struct alignas(64) G1Packed {
  uint32_t w[16];
};

// The codec of a canonical residue (limbs 0..7 in [0, 2^29), limb 8 < 2^22): the plain 256-bit little-endian integer.
// Every limb straddles two words: a 64-bit funnel shift + mask each.  (Measured and removed, round 3: limbs 0..7 in the low
// 29 bits of the words and limb 8 spread over their top bits -- fewer instructions, k_accumulate 1.11 ms either way:
// profiles/r03_ab_combine_pack.txt, git tag exp/pack-spread.)
SNARKV_HD void fq29_pack256(const Fq29& a, uint32_t w[8]) {  // a: canonical residue, limbs in [0, 2^29)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int bit = 32 * j;
    int i = bit / 29, sh = bit % 29;
    uint64_t v = (uint64_t)(uint32_t)a.v[i];
    if (i + 1 < 9) v |= (uint64_t)(uint32_t)a.v[i + 1] << 29;
    if (i + 2 < 9 && sh + 32 > 58) v |= (uint64_t)(uint32_t)a.v[i + 2] << 58;
    w[j] = (uint32_t)(v >> sh);
  }
}

SNARKV_HD Fq29 fq29_unpack256(const uint32_t w[8]) {
  Fq29 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    int bit = 29 * i;
    int word = bit >> 5, sh = bit & 31;
    uint64_t v = w[word];
    if (word + 1 < 8) v |= (uint64_t)w[word + 1] << 32;
    a.v[i] = (int32_t)((uint32_t)(v >> sh) & (uint32_t)kMask29);
  }
  return a;
}

SNARKV_HD G1Packed g1a29_pack(const G1Affine29& p) {
  G1Packed r;
  fq29_pack256(p.x, r.w);
  fq29_pack256(p.y, r.w + 8);
  return r;
}

SNARKV_HD G1Affine29 g1a29_unpack(const G1Packed& k) {
  G1Affine29 r;
  r.x = fq29_unpack256(k.w);
  r.y = fq29_unpack256(k.w + 8);
  return r;
}

SNARKV_HD bool g1a29_is_identity(const G1Affine29& p) { return fq29_limbs_all_zero(p.x) && fq29_limbs_all_zero(p.y); }

SNARKV_HD G1Xyzz29 xyzz29_identity() {
  G1Xyzz29 r;
  r.x = fq29_zero();
  r.y = fq29_zero();
  r.zz = fq29_zero();
  r.zzz = fq29_zero();
  return r;
}

// stored-identity test (exact zero limbs)
SNARKV_HD bool xyzz29_is_identity(const G1Xyzz29& p) { return fq29_limbs_all_zero(p.zz); }

// end-of-run test: ZZ = 0 (mod p) <=> an exceptional case happened (or the
// true result is the identity)
SNARKV_HD bool xyzz29_is_degenerate(const G1Xyzz29& p) { return fq29_is_zero_mod_p(p.zz); }

SNARKV_HD G1Xyzz29 xyzz29_from_affine(const G1Affine29& p) {
  if (g1a29_is_identity(p)) return xyzz29_identity();
  G1Xyzz29 r;
  r.x = p.x;
  r.y = p.y;
  r.zz = fq29_one();
  r.zzz = fq29_one();
  return r;
}

// ---- shared tail of madd / add: given U1 (=X1 scaled), S1, P, R (normalised) --
//   X3 = R^2 - PPP - 2Q,  Y3 = R (Q - X3) - S1 PPP
SNARKV_HD void xyzz29_finish(G1Xyzz29& acc, const Fq29& u1, const Fq29& s1, const Fq29& pn, const Fq29& rn,
                             Fq29& pp, Fq29& ppp) {
  pp = fq29_sqr(pn);
  ppp = fq29_mul(pn, pp);
  Fq29 q = fq29_mul(u1, pp);
  Fq29 rr = fq29_sqr(rn);
  Fq29 x3 = fq29_norm(fq29_sub(fq29_sub(rr, ppp), fq29_dbl(q)));  // limbs before norm in (-3*2^29, 2^29)
  Fq29 t = fq29_sub(q, x3);                                         // |limb| < 2^29
  Fq29 y3 = fq29_mul2(rn, t, fq29_neg(s1), ppp);                    // one Montgomery reduction for both products
  acc.x = x3;
  acc.y = y3;
}

// acc += P (affine, non-identity), acc non-identity.  madd-2008-s, 8M + 2S.
SNARKV_HD void xyzz29_madd_fast(G1Xyzz29& acc, const G1Affine29& p) {
  Fq29 u2 = fq29_mul(p.x, acc.zz);
  Fq29 s2 = fq29_mul(p.y, acc.zzz);
  Fq29 pn = fq29_norm(fq29_sub(u2, acc.x));  // limbs (-2^29, 2^30) -> norm
  Fq29 rn = fq29_norm(fq29_sub(s2, acc.y));
  Fq29 pp, ppp;
  Fq29 x1 = acc.x, y1 = acc.y;
  xyzz29_finish(acc, x1, y1, pn, rn, pp, ppp);
  acc.zz = fq29_mul(acc.zz, pp);
  acc.zzz = fq29_mul(acc.zzz, ppp);
}

// acc += b, both non-identity.  add-2008-s, 12M + 2S.
SNARKV_HD void xyzz29_add_fast(G1Xyzz29& acc, const G1Xyzz29& b) {
  Fq29 u1 = fq29_mul(acc.x, b.zz);
  Fq29 u2 = fq29_mul(b.x, acc.zz);
  Fq29 s1 = fq29_mul(acc.y, b.zzz);
  Fq29 s2 = fq29_mul(b.y, acc.zzz);
  Fq29 pn = fq29_norm(fq29_sub(u2, u1));
  Fq29 rn = fq29_norm(fq29_sub(s2, s1));
  Fq29 pp, ppp;
  xyzz29_finish(acc, u1, s1, pn, rn, pp, ppp);
  acc.zz = fq29_mul(fq29_mul(acc.zz, b.zz), pp);
  acc.zzz = fq29_mul(fq29_mul(acc.zzz, b.zzz), ppp);
}

// 2*P (dbl-2008-s-1, a = 0), P non-identity.
SNARKV_HD G1Xyzz29 xyzz29_double(const G1Xyzz29& p) {
  G1Xyzz29 r;
  Fq29 u = fq29_norm(fq29_dbl(p.y));
  Fq29 v = fq29_sqr(u);
  Fq29 w = fq29_mul(u, v);
  Fq29 s = fq29_mul(p.x, v);
  Fq29 xx = fq29_sqr(p.x);  // p.x is carry-normalised by invariant
  Fq29 m = fq29_norm(fq29_add(fq29_dbl(xx), xx));
  r.x = fq29_norm(fq29_sub(fq29_sqr(m), fq29_dbl(s)));
  r.y = fq29_mul2(m, fq29_sub(s, r.x), fq29_neg(w), p.y);
  r.zz = fq29_mul(v, p.zz);
  r.zzz = fq29_mul(w, p.zzz);
  return r;
}

// 2^n * P for a long run of doublings: Jacobian doubling (dbl-2009-l, a = 0)
// costs 2M + 5S against 6M + 3S for the XYZZ form.  (X, Y, ZZ, ZZZ) ->
// Jacobian (X*ZZ, Y*ZZZ, ZZ) [x = X'/Z'^2 with Z' = ZZ, since ZZ^3 = ZZZ^2] and
// back as (X', Y', Z'^2, Z'^3).  P non-identity.
// Bounds: x, y, z stay carry-normalised; every value passes through a square
// or product each round, so magnitudes stay below ~12p and every product
// result below ~2p (|a*b| / 2^261 < 1.1 p for |a|,|b| < 14p).
// one Jacobian doubling in place (dbl-2009-l, a = 0; 2M + 5S); x, y, z carry-normalised
SNARKV_HD void jac29_double(Fq29& x, Fq29& y, Fq29& z) {
  Fq29 a = fq29_sqr(x);
  Fq29 b = fq29_sqr(y);
  Fq29 c = fq29_sqr(b);
  Fq29 xb = fq29_norm(fq29_add(x, b));
  // D = 2((X+B)^2 - A - C): limbs in (-2^31+4, 2^30) before the carry pass
  Fq29 d = fq29_norm(fq29_dbl(fq29_sub(fq29_sub(fq29_sqr(xb), a), c)));
  Fq29 e = fq29_norm(fq29_add(fq29_dbl(a), a));                    // E = 3A
  Fq29 x3 = fq29_norm(fq29_sub(fq29_sqr(e), fq29_dbl(d)));         // F - 2D
  Fq29 c8 = fq29_dbl(fq29_norm(fq29_dbl(fq29_dbl(c))));            // 8C, limbs < 2^30
  Fq29 y3 = fq29_norm(fq29_sub(fq29_mul(e, fq29_sub(d, x3)), c8));  // E(D - X3) - 8C
  Fq29 z3 = fq29_norm(fq29_dbl(fq29_mul(y, z)));                    // 2YZ
  x = x3;
  y = y3;
  z = z3;
}

// Jacobian (X, Y, Z) -> (X, Y, Z^2, Z^3)
SNARKV_HD G1Xyzz29 jac29_to_xyzz(const Fq29& x, const Fq29& y, const Fq29& z) {
  G1Xyzz29 r;
  Fq29 zz = fq29_sqr(z);
  r.x = x;
  r.y = y;
  r.zz = zz;
  r.zzz = fq29_mul(z, zz);
  return r;
}

#if defined(__HIPCC__)
// One Jacobian doubling (dbl-2009-l, a = 0) of a point held UNIFORMLY by the four lanes of an aligned quad (q = lane & 3),
// its 7 products spread over 3 lanes in 3 dependent levels  { X^2, Y^2, Y Z } -> { B^2, (X+B)^2, (3A)^2 } ->
// { E (D - X3) }  and exchanged by quad-broadcast DPP.  A doubling chain is one wavefront's critical path (a lone
// wavefront issues one VALU instruction per ~4.6 cycles whatever its ILP), so depth 3 instead of 7 products per doubling
// is what counts: the 2^(c w) shift chains of the Pippenger tail and the chunk-base chains of the segmented small MSMs.
// Same dataflow and carry bounds as jac29_double.
template <int CTRL>
__device__ __forceinline__ Fq29 fq29_dpp(const Fq29& a) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = __builtin_amdgcn_update_dpp(0, a.v[i], CTRL, 0xF, 0xF, true);
  return r;
}
__device__ __forceinline__ Fq29 fq29_sel(bool c, const Fq29& a, const Fq29& b) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
__device__ __forceinline__ void jac29_double_quad(Fq29& x, Fq29& y, Fq29& z, uint32_t q) {
  Fq29 p1 = fq29_mul(fq29_sel(q == 0, x, y), fq29_sel(q == 0, x, fq29_sel(q == 1, y, z)));
  Fq29 a = fq29_dpp<0x00>(p1), b = fq29_dpp<0x55>(p1), yz = fq29_dpp<0xAA>(p1);  // quad_perm broadcasts of lane 0 / 1 / 2
  Fq29 xb = fq29_norm(fq29_add(x, b));
  Fq29 e = fq29_norm(fq29_add(fq29_dbl(a), a));  // E = 3A
  Fq29 p2 = fq29_sqr(fq29_sel(q == 0, b, fq29_sel(q == 1, xb, e)));
  Fq29 c = fq29_dpp<0x00>(p2), xb2 = fq29_dpp<0x55>(p2), f = fq29_dpp<0xAA>(p2);
  Fq29 d = fq29_norm(fq29_dbl(fq29_sub(fq29_sub(xb2, a), c)));  // D = 2((X+B)^2 - A - C)
  Fq29 x3 = fq29_norm(fq29_sub(f, fq29_dbl(d)));                // F - 2D
  Fq29 c8 = fq29_dbl(fq29_norm(fq29_dbl(fq29_dbl(c))));         // 8C
  y = fq29_norm(fq29_sub(fq29_mul(e, fq29_sub(d, x3)), c8));    // E(D - X3) - 8C
  z = fq29_norm(fq29_dbl(yz));                                  // 2YZ
  x = x3;
}
#endif

SNARKV_HD G1Xyzz29 xyzz29_double_n(const G1Xyzz29& p, int n) {
  if (n <= 0) return p;
  Fq29 x = fq29_mul(p.x, p.zz);
  Fq29 y = fq29_mul(fq29_norm(p.y), p.zzz);
  Fq29 z = p.zz;
  for (int k = 0; k < n; ++k) jac29_double(x, y, z);
  return jac29_to_xyzz(x, y, z);
}

SNARKV_HD G1Xyzz29 xyzz29_double_affine(const G1Affine29& p) {
  G1Xyzz29 t;
  t.x = p.x;
  t.y = p.y;
  t.zz = fq29_one();
  t.zzz = fq29_one();
  return xyzz29_double(t);
}

// ---- careful flavours: explicit exceptional cases --------------------------
SNARKV_HD void xyzz29_madd_careful(G1Xyzz29& acc, const G1Affine29& p) {
  if (g1a29_is_identity(p)) return;
  if (xyzz29_is_identity(acc)) {
    acc = xyzz29_from_affine(p);
    return;
  }
  Fq29 u2 = fq29_mul(p.x, acc.zz);
  Fq29 s2 = fq29_mul(p.y, acc.zzz);
  Fq29 pn = fq29_norm(fq29_sub(u2, acc.x));
  Fq29 rn = fq29_norm(fq29_sub(s2, acc.y));
  if (fq29_is_zero_mod_p(pn)) {
    if (fq29_is_zero_mod_p(rn)) {
      acc = xyzz29_double_affine(p);
    } else {
      acc = xyzz29_identity();
    }
    return;
  }
  Fq29 pp, ppp;
  Fq29 x1 = acc.x, y1 = acc.y;
  xyzz29_finish(acc, x1, y1, pn, rn, pp, ppp);
  acc.zz = fq29_mul(acc.zz, pp);
  acc.zzz = fq29_mul(acc.zzz, ppp);
}

SNARKV_HD void xyzz29_add_careful(G1Xyzz29& acc, const G1Xyzz29& b) {
  if (xyzz29_is_identity(b)) return;
  if (xyzz29_is_identity(acc)) {
    acc = b;
    return;
  }
  Fq29 u1 = fq29_mul(acc.x, b.zz);
  Fq29 u2 = fq29_mul(b.x, acc.zz);
  Fq29 s1 = fq29_mul(acc.y, b.zzz);
  Fq29 s2 = fq29_mul(b.y, acc.zzz);
  Fq29 pn = fq29_norm(fq29_sub(u2, u1));
  Fq29 rn = fq29_norm(fq29_sub(s2, s1));
  if (fq29_is_zero_mod_p(pn)) {
    if (fq29_is_zero_mod_p(rn)) {
      acc = xyzz29_double(acc);
    } else {
      acc = xyzz29_identity();
    }
    return;
  }
  Fq29 pp, ppp;
  xyzz29_finish(acc, u1, s1, pn, rn, pp, ppp);
  acc.zz = fq29_mul(fq29_mul(acc.zz, b.zz), pp);
  acc.zzz = fq29_mul(fq29_mul(acc.zzz, b.zzz), ppp);
}

// acc += b where either may be the stored identity (exact zero ZZ); fast
// formulas otherwise.  A fast addition that meets an exceptional case leaves
// ZZ = 0 (mod p), i.e. the integer 0 or p.  The integer 0 would later pass for
// the stored identity, so it is caught HERE (`bad`, sticky); the value p keeps
// its non-zero limbs, poisons everything it is added to, and is caught by the
// caller's final `xyzz29_is_degenerate`.
SNARKV_HD void xyzz29_add_skipid_fast(G1Xyzz29& acc, const G1Xyzz29& b, bool& bad) {
  if (xyzz29_is_identity(b)) return;
  if (xyzz29_is_identity(acc)) {
    acc = b;
    return;
  }
  xyzz29_add_fast(acc, b);
  bad = bad || fq29_limbs_all_zero(acc.zz);
}

// Make a result storable: a degenerate ZZ (= 0 mod p but non-zero limbs) must
// never reach memory as a non-identity.  Callers use it after a careful run.
SNARKV_HD G1Xyzz29 xyzz29_sanitize(const G1Xyzz29& p) {
  if (xyzz29_is_identity(p)) return xyzz29_identity();
  return p;
}

// `to_affine`: canonical Montgomery affine; identity -> (0,0)
SNARKV_HD G1Affine29 xyzz29_to_affine(const G1Xyzz29& p) {
  G1Affine29 r;
  if (xyzz29_is_identity(p) || xyzz29_is_degenerate(p)) {
    r.x = fq29_zero();
    r.y = fq29_zero();
    return r;
  }
  Fq29 zn = fq29_norm(fq29_mul(p.zz, p.zzz));
  Fq29 i = fq29_inv(zn);
  Fq29 izz = fq29_mul(i, p.zzz);
  Fq29 izzz = fq29_mul(i, p.zz);
  r.x = fq29_mul(p.x, izz);
  r.y = fq29_mul(fq29_norm(p.y), izzz);
  return r;
}

// 16 words x || y -> Montgomery affine point; the identity is 64 zero bytes in both encodings (halo2curves' (0, 0))
SNARKV_HD G1Affine29 g1a29_from_words(const uint32_t w[16], bool mont) {
  G1Affine29 r;
  bool id = true;
#pragma unroll
  for (int i = 0; i < 16; ++i) id = id && (w[i] == 0);
  if (id) {
    r.x = fq29_zero();
    r.y = fq29_zero();
    return r;
  }
  // canonical residues so that the stored affine point has limbs in [0, 2^29)
  r.x = fq29_canon_of_product(fq29_from_words(w, mont));  // fq29_from_words ends in a product
  r.y = fq29_canon_of_product(fq29_from_words(w + 8, mont));
  return r;
}
SNARKV_HD G1Affine29 g1a29_from_canonical(const uint32_t w[16]) { return g1a29_from_words(w, false); }

SNARKV_HD void g1a29_to_words(const G1Affine29& p, uint32_t w[16], bool mont) {
  fq29_to_words(p.x, w, mont);
  fq29_to_words(p.y, w + 8, mont);
}
SNARKV_HD void g1a29_to_canonical(const G1Affine29& p, uint32_t w[16]) { g1a29_to_words(p, w, false); }

SNARKV_HD G1Affine29 g1a29_neg(const G1Affine29& p) {
  G1Affine29 r;
  r.x = p.x;
  r.y = fq29_neg(p.y);
  return r;
}

// k*P, 256-step double-and-add (`*base * scalar`, reference native.rs:67).
// Canonical scalars (< r) never meet an exceptional case after the first
// addition; non-canonical ones are caught by the degenerate check + careful redo.
template <bool CAREFUL>
SNARKV_HD G1Xyzz29 g1_29_scalar_mul(const G1Affine29& p, const uint32_t k[8]) {
  G1Xyzz29 acc = xyzz29_identity();
  if (g1a29_is_identity(p)) return acc;
  bool started = false;
  for (int i = 7; i >= 0; --i) {
    uint32_t w = k[i];
    for (int b = 31; b >= 0; --b) {
      if (started) {
        if (CAREFUL) {
          if (!xyzz29_is_identity(acc)) acc = xyzz29_double(acc);
        } else {
          acc = xyzz29_double(acc);
        }
      }
      if ((w >> b) & 1u) {
        if (!started) {
          acc = xyzz29_from_affine(p);
          started = true;
        } else if (CAREFUL) {
          xyzz29_madd_careful(acc, p);
        } else {
          xyzz29_madd_fast(acc, p);
        }
      }
    }
  }
  return acc;
}

}  // namespace snarkv
