// Keccak / EVM transcript on the native loader (first half).
//
//   reference                                                   here
//   `EvmTranscript<C, NativeLoader, S, Vec<u8>>`                 EvmTranscript
//     snark-verifier/src/system/halo2/transcript/evm.rs:134-268    (squeeze / common / read)
//     evm.rs:373-398 (halo2 `TranscriptWrite`)                      write_ec_point / write_scalar
//   `u256_to_fe`  loader/evm/util.rs:61-67                        fr_from_be_mod_r
//   `sha3::Keccak256` (external crate, evm.rs:16)                 keccak256 (Keccak-f[1600], rate 136, pad 0x01)
//
// Wire format: every scalar and coordinate is 32 bytes BIG-endian (the EVM
// word order, evm.rs:207-210,219,239,253); challenges are the Keccak-256 of the
// absorbed bytes read as a big-endian integer mod r; after a squeeze the buffer
// holds the 32 hash bytes, and a second squeeze with nothing absorbed in
// between hashes them with a trailing 0x01 (evm.rs:188-193).
// Host-only: transcripts are sequential hashing of a few KB; nothing here
// reaches the device.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "pcs.hpp"

namespace snarkv_host {

namespace keccak {

inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void f1600(uint64_t a[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
      0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
      0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
      0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
      0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  for (int rnd = 0; rnd < 24; ++rnd) {
    uint64_t c[5], b[25];
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; ++x) {
      uint64_t d = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
      for (int y = 0; y < 25; y += 5) a[x + y] ^= d;
    }
    // rho and pi: B[y][2x+3y] = rot(A[x][y], (t+1)(t+2)/2) along the orbit of (1, 0)
    b[0] = a[0];
    int x = 1, y = 0;
    for (int t = 0; t < 24; ++t) {
      int r = ((t + 1) * (t + 2) / 2) % 64;
      int nx = y, ny = (2 * x + 3 * y) % 5;
      b[nx + 5 * ny] = rotl64(a[x + 5 * y], r);
      x = nx;
      y = ny;
    }
    for (int yy = 0; yy < 25; yy += 5)
      for (int xx = 0; xx < 5; ++xx) a[xx + yy] = b[xx + yy] ^ (~b[(xx + 1) % 5 + yy] & b[(xx + 2) % 5 + yy]);
    a[0] ^= RC[rnd];
  }
}

// Keccak-256 (original padding 0x01 ... 0x80), little-endian lanes
inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  const size_t rate = 136;
  uint64_t st[25];
  memset(st, 0, sizeof st);
  auto absorb_block = [&](const uint8_t* blk) {
    for (size_t i = 0; i < rate / 8; ++i) {
      uint64_t w = 0;
      for (int k = 7; k >= 0; --k) w = (w << 8) | blk[8 * i + k];
      st[i] ^= w;
    }
    f1600(st);
  };
  while (len >= rate) {
    absorb_block(data);
    data += rate;
    len -= rate;
  }
  uint8_t last[136];
  memset(last, 0, rate);
  memcpy(last, data, len);
  last[len] ^= 0x01;
  last[rate - 1] ^= 0x80;
  absorb_block(last);
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 8; ++k) out[8 * i + k] = (uint8_t)(st[i] >> (8 * k));
}

}  // namespace keccak

// ---- BN254 base field checks the transcript needs on the host -----------------
namespace fq_host {
static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull,
                                  0x30644e72e131a029ull};
inline bool lt_p(const uint64_t a[4]) {
  for (int i = 3; i >= 0; --i)
    if (a[i] != P[i]) return a[i] < P[i];
  return false;
}
inline void add_mod(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {  // a, b < p
  unsigned __int128 c = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; ++i) {
    c += (unsigned __int128)a[i] + b[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  if (!lt_p(t)) {  // p < 2^254: no carry out of 256 bits
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 x = (unsigned __int128)t[i] - P[i] - (uint64_t)br;
      t[i] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
  }
  memcpy(r, t, 32);
}
// a*b mod p by double-and-add (a few dozen points per proof: speed is irrelevant, obviousness is not)
inline void mul_mod(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t acc[4] = {0, 0, 0, 0};
  for (int i = 255; i >= 0; --i) {
    add_mod(acc, acc, acc);
    if ((b[i >> 6] >> (i & 63)) & 1) add_mod(acc, acc, a);
  }
  memcpy(r, acc, 32);
}
// `C::from_xy`: canonical coordinates and y^2 = x^3 + 3 (the point at infinity has no coordinates)
inline bool g1_from_xy_ok(const uint8_t x_le[32], const uint8_t y_le[32]) {
  uint64_t x[4], y[4];
  memcpy(x, x_le, 32);
  memcpy(y, y_le, 32);
  if (!lt_p(x) || !lt_p(y)) return false;
  uint64_t y2[4], x2[4], x3[4], three[4] = {3, 0, 0, 0}, rhs[4];
  mul_mod(y2, y, y);
  mul_mod(x2, x, x);
  mul_mod(x3, x2, x);
  add_mod(rhs, x3, three);
  return memcmp(y2, rhs, 32) == 0;
}
}  // namespace fq_host

// `u256_to_fe(U256::from_be_bytes(hash))`, loader/evm/util.rs:61-67
inline Fr fr_from_be_mod_r(const uint8_t be[32]) {
  uint64_t w[4];
  for (int i = 0; i < 4; ++i) {
    uint64_t v = 0;
    for (int k = 0; k < 8; ++k) v = (v << 8) | be[8 * (3 - i) + k];
    w[i] = v;
  }
  auto ge_r = [&]() {
    for (int i = 3; i >= 0; --i)
      if (w[i] != Fr::MOD[i]) return w[i] > Fr::MOD[i];
    return true;
  };
  while (ge_r()) {  // 2^256 / r < 6
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 x = (unsigned __int128)w[i] - Fr::MOD[i] - (uint64_t)br;
      w[i] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
  }
  uint8_t le[32];
  memcpy(le, w, 32);
  Fr out;
  Fr::from_bytes(le, &out);
  return out;
}

class EvmTranscript : public Transcript {
 public:
  EvmTranscript() = default;
  explicit EvmTranscript(std::vector<uint8_t> proof) : stream_(std::move(proof)) {}

  // evm.rs:184-198
  Fr squeeze_challenge() override {
    std::vector<uint8_t> data = buf_;
    if (buf_.size() == 0x20) data.push_back(1);
    uint8_t h[32];
    keccak::keccak256(data.data(), data.size(), h);
    buf_.assign(h, h + 32);
    return fr_from_be_mod_r(h);
  }

  // evm.rs:200-216: the point at infinity has no coordinates -> Error::Transcript
  Error common_ec_point(const G1Affine& p) override {
    if (p.is_identity()) return Error{Error::Transcript, "Invalid elliptic curve point"};
    push_be(p.b);
    push_be(p.b + 32);
    return Error{};
  }

  // evm.rs:218-222
  Error common_scalar(const Fr& s) override {
    uint8_t le[32];
    s.to_bytes(le);
    push_be(le);
    return Error{};
  }

  // evm.rs:231-245
  Result<Fr> read_scalar() override {
    uint8_t le[32];
    if (!read_be(le)) return Result<Fr>::Err(Error{Error::Transcript, "failed to fill whole buffer"});
    Fr s;
    if (!Fr::from_bytes(le, &s)) return Result<Fr>::Err(Error{Error::Transcript, "Invalid scalar encoding in proof"});
    common_scalar(s);
    return Result<Fr>::Ok(s);
  }

  // evm.rs:247-268
  Result<G1Affine> read_ec_point() override {
    G1Affine p;
    if (!read_be(p.b) || !read_be(p.b + 32))
      return Result<G1Affine>::Err(Error{Error::Transcript, "failed to fill whole buffer"});
    if (!fq_host::g1_from_xy_ok(p.b, p.b + 32))
      return Result<G1Affine>::Err(Error{Error::Transcript, "Invalid elliptic curve point encoding in proof"});
    common_ec_point(p);
    return Result<G1Affine>::Ok(p);
  }

  // evm.rs:373-388
  Error write_ec_point(const G1Affine& p) override {
    Error e = common_ec_point(p);
    if (!e.ok()) return Error{Error::Transcript, "Cannot write points at infinity to the transcript"};
    append_be(stream_, p.b);
    append_be(stream_, p.b + 32);
    return Error{};
  }

  // evm.rs:390-396
  Error write_scalar(const Fr& s) override {
    common_scalar(s);
    uint8_t le[32];
    s.to_bytes(le);
    append_be(stream_, le);
    return Error{};
  }

  const std::vector<uint8_t>& stream() const { return stream_; }
  std::vector<uint8_t> finalize() { return std::move(stream_); }  // evm.rs:283-286
  size_t remaining() const { return stream_.size() - pos_; }

 private:
  static void append_be(std::vector<uint8_t>& v, const uint8_t le[32]) {
    for (int i = 31; i >= 0; --i) v.push_back(le[i]);
  }
  void push_be(const uint8_t le[32]) { append_be(buf_, le); }
  bool read_be(uint8_t le[32]) {
    if (pos_ + 32 > stream_.size()) return false;
    for (int i = 0; i < 32; ++i) le[i] = stream_[pos_ + 31 - i];
    pos_ += 32;
    return true;
  }
  std::vector<uint8_t> stream_;
  size_t pos_ = 0;
  std::vector<uint8_t> buf_;
};

}  // namespace snarkv_host
