// halo2's Blake2b transcript on the native loader -- `halo2_proofs::transcript::{Blake2bRead,
// Blake2bWrite}` with `Challenge255`, the transcript the reference's IPA tests use on pallas
// (snark-verifier/src/pcs/ipa.rs:438-440, pcs/ipa/accumulation.rs:244-247,
// system/halo2/test/ipa/native.rs:11-12).  The crate is external (not under /root/reference):
// restated from its published definition, as oracle/transcript.py `Blake2bTranscript` --
//   state   BLAKE2b-512, personalisation "Halo2-Transcript"
//   prefix  0x00 before a challenge, 0x01 before a point (x, y: 32-byte LE each), 0x02 before a scalar
//   squeeze digest of a COPY of the state, read little-endian, reduced mod r (`from_uniform_bytes`)
//   points  travel compressed: x with the parity of y in bit 255; the identity is never written
// BLAKE2b itself follows RFC 7693.  PARITY UNPINNED (no fixtures in the reference).
//
// Pasta flavour of the mirror only (-DSNARKV_HOST_PALLAS): decompression needs the base field of the
// curve, here pallas' p = 2^254 + 45560315531419706090280762371685220353 (Tonelli-Shanks: p = 1 mod 2^32).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "pcs.hpp"

namespace snarkv_host {

// ---- BLAKE2b (RFC 7693), unkeyed, with personalisation; incremental, copyable state ----------
class Blake2b {
 public:
  explicit Blake2b(size_t outlen, const char* person16) : outlen_(outlen) {
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                   0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                   0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    memcpy(h_, IV, sizeof h_);
    h_[0] ^= 0x01010000ull ^ (uint64_t)outlen;  // digest length, key length 0, fanout 1, depth 1
    uint64_t p[2] = {0, 0};
    memcpy(p, person16, strnlen(person16, 16));
    h_[6] ^= p[0];  // parameter block bytes 48..63 = personalisation
    h_[7] ^= p[1];
  }
  void update(const uint8_t* in, size_t len) {
    while (len > 0) {
      if (fill_ == 128) {  // the buffer is only compressed once more input arrives (the last block is special)
        t_ += 128;
        compress(false);
        fill_ = 0;
      }
      size_t take = std::min(len, (size_t)128 - fill_);
      memcpy(buf_ + fill_, in, take);
      fill_ += take;
      in += take;
      len -= take;
    }
  }
  // digest of everything absorbed so far; the object can keep absorbing (works on a copy)
  void digest(uint8_t* out) const {
    Blake2b c = *this;
    c.t_ += c.fill_;
    memset(c.buf_ + c.fill_, 0, 128 - c.fill_);
    c.compress(true);
    memcpy(out, c.h_, outlen_);
  }

 private:
  static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void compress(bool last) {
    static const uint8_t S[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                   0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                   0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    uint64_t m[16], v[16];
    memcpy(m, buf_, 128);
    for (int i = 0; i < 8; ++i) {
      v[i] = h_[i];
      v[i + 8] = IV[i];
    }
    v[12] ^= t_;  // message length fits 64 bits here
    if (last) v[14] = ~v[14];
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
      v[a] = v[a] + v[b] + x;
      v[d] = rotr(v[d] ^ v[a], 32);
      v[c] = v[c] + v[d];
      v[b] = rotr(v[b] ^ v[c], 24);
      v[a] = v[a] + v[b] + y;
      v[d] = rotr(v[d] ^ v[a], 16);
      v[c] = v[c] + v[d];
      v[b] = rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; ++r) {
      const uint8_t* s = S[r];
      G(0, 4, 8, 12, m[s[0]], m[s[1]]);
      G(1, 5, 9, 13, m[s[2]], m[s[3]]);
      G(2, 6, 10, 14, m[s[4]], m[s[5]]);
      G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      G(0, 5, 10, 15, m[s[8]], m[s[9]]);
      G(1, 6, 11, 12, m[s[10]], m[s[11]]);
      G(2, 7, 8, 13, m[s[12]], m[s[13]]);
      G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[i + 8];
  }
  uint64_t h_[8];
  uint64_t t_ = 0;
  uint8_t buf_[128] = {0};
  size_t fill_ = 0;
  size_t outlen_;
};

// ---- pallas base field, just enough for `C::from_bytes`: y = sqrt(x^3 + 5) with the sign bit ----
namespace pallas_fp {
static constexpr uint64_t P[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0000000000000000ull,
                                  0x4000000000000000ull};
static constexpr uint64_t INV = 0x992d30ecffffffffull;  // -p^-1 mod 2^64
static constexpr uint64_t ONE_M[4] = {0x34786d38fffffffdull, 0x992c350be41914adull, 0xffffffffffffffffull,
                                      0x3fffffffffffffffull};  // 2^256 mod p
static constexpr uint64_t R2[4] = {0x8c78ecb30000000full, 0xd7d30dbd8b0de0e7ull, 0x7797a99bc3c95d18ull,
                                   0x096d41af7b9cb714ull};  // 2^512 mod p
struct Fp {
  uint64_t v[4];  // Montgomery form
  bool operator==(const Fp& o) const { return memcmp(v, o.v, 32) == 0; }
};
inline bool lt_p(const uint64_t a[4]) {
  for (int i = 3; i >= 0; --i)
    if (a[i] != P[i]) return a[i] < P[i];
  return false;
}
inline void sub_p(uint64_t t[4]) {
  unsigned __int128 br = 0;
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 x = (unsigned __int128)t[i] - P[i] - (uint64_t)br;
    t[i] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
}
inline Fp mul(const Fp& a, const Fp& b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (unsigned __int128)a.v[i] * b.v[j] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * INV;
    c = ((unsigned __int128)m * P[0] + t[0]) >> 64;
    for (int j = 1; j < 4; ++j) {
      c += (unsigned __int128)m * P[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || !lt_p(t)) sub_p(t);
  Fp r;
  memcpy(r.v, t, 32);
  return r;
}
inline Fp add(const Fp& a, const Fp& b) {
  uint64_t t[4];
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (unsigned __int128)a.v[i] + b.v[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  if (!lt_p(t)) sub_p(t);  // 2p < 2^256
  Fp r;
  memcpy(r.v, t, 32);
  return r;
}
inline Fp one() {
  Fp r;
  memcpy(r.v, ONE_M, 32);
  return r;
}
inline Fp from_canonical(const uint64_t w[4]) {
  Fp a, r2;
  memcpy(a.v, w, 32);
  memcpy(r2.v, R2, 32);
  return mul(a, r2);
}
inline void to_canonical(const Fp& a, uint64_t w[4]) {
  Fp raw{{1, 0, 0, 0}};
  Fp t = mul(a, raw);
  memcpy(w, t.v, 32);
}
inline Fp pow(const Fp& a, const uint64_t e[4]) {
  Fp r = one();
  for (int i = 3; i >= 0; --i)
    for (int b = 63; b >= 0; --b) {
      r = mul(r, r);
      if ((e[i] >> b) & 1) r = mul(r, a);
    }
  return r;
}
// Tonelli-Shanks with p - 1 = 2^32 * t.  false if `a` is not a square.
inline bool sqrt(const Fp& a, Fp* out) {
  const Fp zero{{0, 0, 0, 0}};
  if (a == zero) {
    *out = zero;
    return true;
  }
  uint64_t t[4], half[4], t1h[4];  // t = (p-1) >> 32, (p-1)/2, (t+1)/2
  for (int i = 0; i < 4; ++i) t[i] = (P[i] >> 32) | (i < 3 ? P[i + 1] << 32 : 0);
  // (the low 32 bits of p are 0x00000001, so p >> 32 == (p - 1) >> 32)
  for (int i = 0; i < 4; ++i) half[i] = (P[i] >> 1) | (i < 3 ? P[i + 1] << 63 : 0);
  {
    uint64_t tp[4] = {t[0] + 1, t[1], t[2], t[3]};  // t is odd: no carry
    for (int i = 0; i < 4; ++i) t1h[i] = (tp[i] >> 1) | (i < 3 ? tp[i + 1] << 63 : 0);
  }
  const Fp minus_one = [] {
    uint64_t w[4] = {P[0] - 1, P[1], P[2], P[3]};
    return from_canonical(w);
  }();
  if (!(pow(a, half) == one())) return false;
  // a non-residue: 5 (the smallest for this p; checked by Euler's criterion at first use)
  uint64_t five[4] = {5, 0, 0, 0};
  Fp z = from_canonical(five);
  if (!(pow(z, half) == minus_one)) return false;
  Fp c = pow(z, t), tt = pow(a, t), r = pow(a, t1h);
  uint32_t m = 32;
  while (!(tt == one())) {
    uint32_t i = 0;
    Fp t2 = tt;
    while (!(t2 == one())) {
      t2 = mul(t2, t2);
      ++i;
    }
    Fp b = c;
    for (uint32_t k = 0; k + i + 1 < m; ++k) b = mul(b, b);
    m = i;
    c = mul(b, b);
    tt = mul(tt, c);
    r = mul(r, b);
  }
  *out = r;
  return true;
}
}  // namespace pallas_fp

// ---- the transcript --------------------------------------------------------------------------
class Blake2bTranscript : public Transcript {
 public:
  Blake2bTranscript() : state_(64, "Halo2-Transcript") {}
  explicit Blake2bTranscript(std::vector<uint8_t> proof) : state_(64, "Halo2-Transcript"), stream_(std::move(proof)) {}

  Fr squeeze_challenge() override {
    const uint8_t prefix = 0;
    state_.update(&prefix, 1);
    uint8_t h[64];
    state_.digest(h);
    // `from_uniform_bytes`: the 512-bit little-endian integer mod r = lo + hi * 2^256
    Fr lo = fr_from_le_mod_r(h), hi = fr_from_le_mod_r(h + 32);
    return lo + hi * two_256();
  }
  Error common_ec_point(const G1Affine& p) override {
    if (p.is_identity()) return Error{Error::Transcript, "cannot write points at infinity to the transcript"};
    const uint8_t prefix = 1;
    state_.update(&prefix, 1);
    state_.update(p.b, 64);
    return {};
  }
  Error common_scalar(const Fr& s) override {
    uint8_t b[33];
    b[0] = 2;
    s.to_bytes(b + 1);
    state_.update(b, 33);
    return {};
  }
  Result<Fr> read_scalar() override {
    uint8_t b[32];
    if (!take(b, 32)) return Result<Fr>::Err(Error{Error::Transcript, "failed to fill whole buffer"});
    Fr s;
    if (!Fr::from_bytes(b, &s)) return Result<Fr>::Err(Error{Error::Transcript, "invalid field element encoding in proof"});
    common_scalar(s);
    return Result<Fr>::Ok(s);
  }
  Result<G1Affine> read_ec_point() override {
    using R = Result<G1Affine>;
    uint8_t b[32];
    if (!take(b, 32)) return R::Err(Error{Error::Transcript, "failed to fill whole buffer"});
    const uint8_t sign = b[31] >> 7;
    b[31] &= 0x7F;
    uint64_t xw[4];
    memcpy(xw, b, 32);
    const Error bad{Error::Transcript, "invalid point encoding in proof"};
    if (!pallas_fp::lt_p(xw)) return R::Err(bad);
    using namespace pallas_fp;
    Fp x = from_canonical(xw);
    uint64_t five[4] = {5, 0, 0, 0};
    Fp rhs = add(mul(mul(x, x), x), from_canonical(five)), y;
    if (!sqrt(rhs, &y)) return R::Err(bad);
    uint64_t yw[4];
    to_canonical(y, yw);
    if ((xw[0] | xw[1] | xw[2] | xw[3]) == 0 && sign == 0) return R::Err(bad);  // the identity's encoding
    if ((yw[0] & 1) != sign) {  // the other root: p - y
      unsigned __int128 br = 0;
      for (int i = 0; i < 4; ++i) {
        unsigned __int128 d = (unsigned __int128)P[i] - yw[i] - (uint64_t)br;
        yw[i] = (uint64_t)d;
        br = (d >> 64) & 1;
      }
    }
    G1Affine p;
    memcpy(p.b, xw, 32);
    memcpy(p.b + 32, yw, 32);
    Error e = common_ec_point(p);
    if (!e.ok()) return R::Err(e);
    return R::Ok(p);
  }
  Error write_ec_point(const G1Affine& p) override {
    Error e = common_ec_point(p);
    if (!e.ok()) return e;
    uint8_t b[32];
    memcpy(b, p.b, 32);
    b[31] |= (uint8_t)((p.b[32] & 1) << 7);
    stream_.insert(stream_.end(), b, b + 32);
    return {};
  }
  Error write_scalar(const Fr& s) override {
    common_scalar(s);
    uint8_t b[32];
    s.to_bytes(b);
    stream_.insert(stream_.end(), b, b + 32);
    return {};
  }
  const std::vector<uint8_t>& finalize() const { return stream_; }

 private:
  static Fr two_256() {
    static const Fr v = [] {
      Fr x = Fr::from_u64(1ull << 32);
      Fr y = x * x;    // 2^64
      Fr z = y * y;    // 2^128
      return z * z;    // 2^256
    }();
    return v;
  }
  // a 256-bit little-endian integer mod r (2^256 / r < 4 for the pasta fields)
  static Fr fr_from_le_mod_r(const uint8_t le[32]) {
    uint64_t w[4];
    memcpy(w, le, 32);
    auto ge_r = [&]() {
      for (int i = 3; i >= 0; --i)
        if (w[i] != Fr::MOD[i]) return w[i] > Fr::MOD[i];
      return true;
    };
    while (ge_r()) {
      unsigned __int128 br = 0;
      for (int i = 0; i < 4; ++i) {
        unsigned __int128 x = (unsigned __int128)w[i] - Fr::MOD[i] - (uint64_t)br;
        w[i] = (uint64_t)x;
        br = (x >> 64) & 1;
      }
    }
    uint8_t b[32];
    memcpy(b, w, 32);
    Fr out;
    Fr::from_bytes(b, &out);
    return out;
  }
  bool take(uint8_t* out, size_t n) {
    if (pos_ + n > stream_.size()) return false;
    memcpy(out, stream_.data() + pos_, n);
    pos_ += n;
    return true;
  }
  Blake2b state_;
  std::vector<uint8_t> stream_;
  size_t pos_ = 0;
};

}  // namespace snarkv_host
