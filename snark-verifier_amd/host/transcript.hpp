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
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "pcs.hpp"
#include "poseidon_ifma.hpp"

namespace snarkv_host {

namespace keccak {

inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

// rho rotation of lane x + 5y and its pi destination, both derived from the
// orbit (x, y) -> (y, 2x + 3y) of (1, 0) with offsets (t+1)(t+2)/2
struct RhoPi {
  int rot[25], dst[25];
  constexpr RhoPi() : rot(), dst() {
    rot[0] = 0;
    dst[0] = 0;
    int x = 1, y = 0;
    for (int t = 0; t < 24; ++t) {
      int nx = y, ny = (2 * x + 3 * y) % 5;
      rot[x + 5 * y] = ((t + 1) * (t + 2) / 2) % 64;
      dst[x + 5 * y] = nx + 5 * ny;
      x = nx;
      y = ny;
    }
  }
};
constexpr RhoPi kTab{};

inline void f1600(uint64_t a[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
      0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
      0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
      0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
      0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  // theta, rho, pi, chi with every lane index and rotation a compile-time constant (kTab is constexpr and the loops
  // are fully unrolled): the 25 lanes stay in registers, ~2x the table-driven loop -- the accumulation transcript of a
  // 1 024-proof job hashes 131 KB
  for (int rnd = 0; rnd < 24; ++rnd) {
    const uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20];
    const uint64_t c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
    const uint64_t c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22];
    const uint64_t c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
    const uint64_t c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
    const uint64_t d[5] = {c4 ^ rotl64(c1, 1), c0 ^ rotl64(c2, 1), c1 ^ rotl64(c3, 1), c2 ^ rotl64(c4, 1),
                           c3 ^ rotl64(c0, 1)};
    uint64_t b[25];
#pragma GCC unroll 25
    for (int i = 0; i < 25; ++i) b[kTab.dst[i]] = rotl64(a[i] ^ d[i % 5], kTab.rot[i]);  // theta, rho, pi
#pragma GCC unroll 5
    for (int y = 0; y < 25; y += 5) {                                                    // chi
      const uint64_t b0 = b[y], b1 = b[y + 1], b2 = b[y + 2], b3 = b[y + 3], b4 = b[y + 4];
      a[y] = b0 ^ (~b1 & b2);
      a[y + 1] = b1 ^ (~b2 & b3);
      a[y + 2] = b2 ^ (~b3 & b4);
      a[y + 3] = b3 ^ (~b4 & b0);
      a[y + 4] = b4 ^ (~b0 & b1);
    }
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
      uint64_t w;
      memcpy(&w, blk + 8 * i, 8);  // little-endian host (x86-64)
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
// a*b mod p on plain integers: two Montgomery products (a b / R, then times R^2 / R)
inline void mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  static constexpr uint64_t INV = 0x87d20782e4866389ull;  // -p^-1 mod 2^64
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (unsigned __int128)a[i] * b[j] + t[j];
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
  if (!lt_p(t)) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 x = (unsigned __int128)t[i] - P[i] - (uint64_t)br;
      t[i] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
  }
  memcpy(r, t, 32);
}
inline void mul_mod(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull,
                                     0x06d89f71cab8351full};  // 2^512 mod p
  uint64_t t[4];
  mont_mul(t, a, b);
  mont_mul(r, t, R2);
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
    if (buf_.size() == 0x20) buf_.push_back(1);  // (the buffer is replaced by the digest below: no copy needed)
    uint8_t h[32];
    keccak::keccak256(buf_.data(), buf_.size(), h);
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
    push_be(le);  // = common_scalar(s): `le` is the canonical encoding of s (from_bytes accepted it)
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
    const size_t o = v.size();
    v.resize(o + 32);
    for (int i = 0; i < 32; ++i) v[o + i] = le[31 - i];
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

// ============================================================================
// Poseidon transcript (second half).
//
//   reference                                                    here
//   `Poseidon<F, L, T, RATE>`  util/hash/poseidon.rs:115-202       Poseidon
//   `PoseidonTranscript<C, NativeLoader, S, T, RATE, R_F, R_P>`    PoseidonTranscript
//      system/halo2/transcript/halo2.rs:170-321
//   `poseidon::Spec::new(r_f, r_p)` (un-vendored crate)            poseidon_spec(): Grain LFSR
//
// The reference runs the optimised round schedule whose tables the external
// crate derives; it is an exact rewriting of the plain permutation, which is
// what is implemented here (constants / Cauchy MDS from the Grain LFSR of the
// Poseidon reference; pinned by the public t = 3 instance, see oracle/transcript.py).
// Unpinned crate internals: `State::default()` = [2^64, 0, ..] and the
// compressed G1 encoding of halo2curves 0.6.0 (bit 7 = identity, bit 6 = y odd).
struct PoseidonSpec {
  int t = 0, r_f = 0, r_p = 0;
  std::vector<Fr> rc;   // (r_f + r_p) * t
  std::vector<Fr> mds;  // t * t, row-major
};

namespace grain {
struct Lfsr {
  bool st[80];
  Lfsr(int n_bits, int t, int r_f, int r_p) {
    int k = 0;
    auto put = [&](unsigned v, int w) {
      for (int i = w - 1; i >= 0; --i) st[k++] = (v >> i) & 1u;
    };
    put(1, 2);   // prime field
    put(0, 4);   // S-box x^alpha
    put((unsigned)n_bits, 12);
    put((unsigned)t, 12);
    put((unsigned)r_f, 10);
    put((unsigned)r_p, 10);
    for (int i = 0; i < 30; ++i) st[k++] = true;
    for (int i = 0; i < 160; ++i) raw();
  }
  bool raw() {
    bool nb = st[62] ^ st[51] ^ st[38] ^ st[23] ^ st[13] ^ st[0];
    memmove(st, st + 1, 79 * sizeof(bool));
    st[79] = nb;
    return nb;
  }
  bool next_bit() {  // self-shrinking output
    for (;;) {
      bool b1 = raw(), b2 = raw();
      if (b1) return b2;
    }
  }
  // n_bits bits, most significant first, as 4 x u64 little-endian words
  void next_int(int n_bits, uint64_t w[4]) {
    w[0] = w[1] = w[2] = w[3] = 0;
    for (int i = n_bits - 1; i >= 0; --i)
      if (next_bit()) w[i >> 6] |= 1ull << (i & 63);
  }
};
inline bool lt_r(const uint64_t w[4]) {
  for (int i = 3; i >= 0; --i)
    if (w[i] != Fr::MOD[i]) return w[i] < Fr::MOD[i];
  return false;
}
inline Fr fr_from_words_mod_r(uint64_t w[4]) {  // w < 2^254 < 2r
  if (!lt_r(w)) {
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
}  // namespace grain

inline const PoseidonSpec& poseidon_spec(int t, int r_f, int r_p) {
  static std::map<std::tuple<int, int, int>, PoseidonSpec> cache;  // node-based: references stay valid
  static std::mutex mu;                                            // transcripts are built from many host threads
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(t, r_f, r_p);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const int n_bits = 254;  // Fr::NUM_BITS
  PoseidonSpec sp;
  sp.t = t;
  sp.r_f = r_f;
  sp.r_p = r_p;
  grain::Lfsr g(n_bits, t, r_f, r_p);
  while ((int)sp.rc.size() < (r_f + r_p) * t) {  // round constants: rejection sampling
    uint64_t w[4];
    g.next_int(n_bits, w);
    if (grain::lt_r(w)) sp.rc.push_back(grain::fr_from_words_mod_r(w));
  }
  for (;;) {  // Cauchy MDS 1 / (x_i + y_j): no rejection, resample until all 2t values are distinct
    std::vector<Fr> v;
    for (int i = 0; i < 2 * t; ++i) {
      uint64_t w[4];
      g.next_int(n_bits, w);
      v.push_back(grain::fr_from_words_mod_r(w));
    }
    bool ok = true;
    for (int i = 0; i < 2 * t && ok; ++i)
      for (int j = i + 1; j < 2 * t && ok; ++j)
        if (v[i] == v[j]) ok = false;
    if (!ok) continue;
    sp.mds.assign((size_t)t * t, Fr::zero());
    for (int i = 0; i < t && ok; ++i)
      for (int j = 0; j < t && ok; ++j) {
        Fr inv;
        if (!(v[i] + v[t + j]).invert(&inv)) ok = false;
        sp.mds[(size_t)i * t + j] = inv;
      }
    if (ok) break;
  }
  return cache.emplace(key, std::move(sp)).first->second;
}

// The plain permutation: R_F/2 full rounds, R_P partial rounds, R_F/2 full rounds, each
// "add constants, S-box, MDS".  Reference semantics; the sponge runs `poseidon_permute`
// below (the optimised schedule), which tests pin against this one.
inline void poseidon_permute_plain(std::vector<Fr>& state, const PoseidonSpec& sp) {
  const int t = sp.t;
  size_t k = 0;
  std::vector<Fr> next((size_t)t);
  for (int rnd = 0; rnd < sp.r_f + sp.r_p; ++rnd) {
    for (int i = 0; i < t; ++i) state[i] = state[i] + sp.rc[k++];
    const bool full = rnd < sp.r_f / 2 || rnd >= sp.r_f / 2 + sp.r_p;
    for (int i = 0; i < (full ? t : 1); ++i) {
      Fr x2 = state[i].square();
      state[i] = x2.square() * state[i];
    }
    for (int i = 0; i < t; ++i) {
      Fr acc = Fr::zero();
      for (int j = 0; j < t; ++j) acc = acc + sp.mds[(size_t)i * t + j] * state[j];
      next[i] = acc;
    }
    state = next;
  }
}

// ---- optimised schedule (the shape the reference runs, poseidon.rs:166-201) ----
// Constants move behind the S-boxes (k_r = M^-1 e_{r+1}; in a partial round only the
// word-0 component stays there, the rest slides in front of that round's S-box, which does
// not touch words 1..), and the partial rounds' dense MDS is factored  M~ = M'' M'  with
// M' = diag(1, m^) commuting with the word-0 S-box: M' merges into the previous round's
// matrix, M'' = [[a, v], [w, I]] is sparse (2t - 1 products instead of t^2).  Derived here
// from the plain spec; the external crate's tables are the same rewriting.
struct PoseidonOpt {
  int t = 0, r_f = 0, r_p = 0;
  std::vector<Fr> pre;                    // t: constants added with the input
  std::vector<std::vector<Fr>> full_k;    // post-S-box constants of the full rounds, in order (last one absent = 0)
  std::vector<Fr> partial_k;              // r_p scalars (word 0)
  std::vector<Fr> mds, pre_sparse;        // t*t
  std::vector<std::vector<Fr>> sparse_row, sparse_col;  // per partial round: row (t), col_hat (t-1)
};

namespace poseidon_detail {
using Mat = std::vector<Fr>;  // row-major n x n
inline Mat mat_mul(const Mat& a, const Mat& b, int n) {
  Mat c((size_t)n * n, Fr::zero());
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k)
      for (int j = 0; j < n; ++j) c[(size_t)i * n + j] = c[(size_t)i * n + j] + a[(size_t)i * n + k] * b[(size_t)k * n + j];
  return c;
}
inline Mat mat_inv(Mat a, int n) {  // Gauss-Jordan; throws if singular
  Mat inv((size_t)n * n, Fr::zero());
  for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = Fr::one();
  for (int col = 0; col < n; ++col) {
    int piv = -1;
    for (int r = col; r < n; ++r)
      if (!a[(size_t)r * n + col].is_zero()) {
        piv = r;
        break;
      }
    if (piv < 0) throw Panic("poseidon: singular matrix in the optimised schedule");
    for (int j = 0; j < n; ++j) {
      std::swap(a[(size_t)piv * n + j], a[(size_t)col * n + j]);
      std::swap(inv[(size_t)piv * n + j], inv[(size_t)col * n + j]);
    }
    Fr pinv;
    a[(size_t)col * n + col].invert(&pinv);
    for (int j = 0; j < n; ++j) {
      a[(size_t)col * n + j] = a[(size_t)col * n + j] * pinv;
      inv[(size_t)col * n + j] = inv[(size_t)col * n + j] * pinv;
    }
    for (int r = 0; r < n; ++r) {
      if (r == col || a[(size_t)r * n + col].is_zero()) continue;
      Fr f = a[(size_t)r * n + col];
      for (int j = 0; j < n; ++j) {
        a[(size_t)r * n + j] = a[(size_t)r * n + j] - f * a[(size_t)col * n + j];
        inv[(size_t)r * n + j] = inv[(size_t)r * n + j] - f * inv[(size_t)col * n + j];
      }
    }
  }
  return inv;
}
inline std::vector<Fr> mat_vec(const Mat& m, const std::vector<Fr>& v, int n) {
  std::vector<Fr> o((size_t)n, Fr::zero());
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) o[i] = o[i] + m[(size_t)i * n + j] * v[j];
  return o;
}
}  // namespace poseidon_detail

inline const PoseidonOpt& poseidon_opt(int t, int r_f, int r_p) {
  static std::map<std::tuple<int, int, int>, PoseidonOpt> cache;
  static std::mutex mu;
  const PoseidonSpec& sp = poseidon_spec(t, r_f, r_p);
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(t, r_f, r_p);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  using namespace poseidon_detail;
  PoseidonOpt o;
  o.t = t;
  o.r_f = r_f;
  o.r_p = r_p;
  o.mds = sp.mds;
  const int h = r_f / 2, R = r_f + r_p;
  auto c = [&](int r) { return std::vector<Fr>(sp.rc.begin() + (size_t)r * t, sp.rc.begin() + (size_t)(r + 1) * t); };
  const Mat minv = mat_inv(sp.mds, t);
  // constants, backwards: e = the pre-S-box vector wanted at round r + 1
  std::vector<std::vector<Fr>> kfull((size_t)R);  // post-S-box vector of full round r
  o.partial_k.assign((size_t)r_p, Fr::zero());
  std::vector<Fr> e = c(R - 1);
  for (int r = R - 2; r >= 0; --r) {
    std::vector<Fr> back = mat_vec(minv, e, t);  // M^-1 e_{r+1}
    const bool partial = r >= h && r < h + r_p;
    e = c(r);
    if (partial) {
      o.partial_k[(size_t)(r - h)] = back[0];
      for (int i = 1; i < t; ++i) e[i] = e[i] + back[i];  // slides in front of this round's (word-0 only) S-box
    } else {
      kfull[(size_t)r] = back;
    }
  }
  o.pre = e;  // = c(0): round 0 is full
  for (int r = 0; r < R - 1; ++r)
    if (!(r >= h && r < h + r_p)) o.full_k.push_back(kfull[(size_t)r]);
  // matrices, backwards over the partial rounds
  Mat cur = sp.mds;
  o.sparse_row.assign((size_t)r_p, {});
  o.sparse_col.assign((size_t)r_p, {});
  for (int r = r_p - 1; r >= 0; --r) {
    const int m = t - 1;
    Mat mhat((size_t)m * m);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) mhat[(size_t)i * m + j] = cur[(size_t)(i + 1) * t + (j + 1)];
    Mat mhat_inv = mat_inv(mhat, m);
    std::vector<Fr> row((size_t)t), col((size_t)m);
    row[0] = cur[0];
    for (int j = 0; j < m; ++j) {  // v = M~[0,1:] m^^-1
      Fr acc = Fr::zero();
      for (int k = 0; k < m; ++k) acc = acc + cur[(size_t)(k + 1)] * mhat_inv[(size_t)k * m + j];
      row[(size_t)j + 1] = acc;
    }
    for (int i = 0; i < m; ++i) col[i] = cur[(size_t)(i + 1) * t];  // w = M~[1:,0]
    o.sparse_row[(size_t)r] = row;
    o.sparse_col[(size_t)r] = col;
    Mat mprime((size_t)t * t, Fr::zero());  // diag(1, m^)
    mprime[0] = Fr::one();
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) mprime[(size_t)(i + 1) * t + (j + 1)] = mhat[(size_t)i * m + j];
    cur = mat_mul(mprime, sp.mds, t);  // the previous round's matrix absorbs M'
  }
  o.pre_sparse = cur;
  return cache.emplace(key, std::move(o)).first->second;
}

// state <- permutation(state): identical values to poseidon_permute_plain, about half the products -- and the products of a
// matrix row are summed as 512-bit integers and reduced once (Fr::Wide): the accumulation transcript of an aggregation is
// ONE sponge (4 m elements, m + 1 DEPENDENT permutations on one thread; 1 025 of them at m = 1 024), so the permutation's
// own speed is what that step costs.  No allocation inside: the state and the scratch row live on the stack.
inline void poseidon_permute(std::vector<Fr>& state_v, const PoseidonSpec& sp) {
  const PoseidonOpt& o = poseidon_opt(sp.t, sp.r_f, sp.r_p);
  const int t = o.t, h = o.r_f / 2;
  constexpr int kMaxT = 16;
  if (t > kMaxT) throw Panic("poseidon: state wider than this implementation's stack arrays");
  Fr state[kMaxT], next[kMaxT];
  for (int i = 0; i < t; ++i) state[i] = state_v[(size_t)i];
  auto sbox = [](const Fr& x) {
    Fr x2 = x.square();
    return x2.square() * x;
  };
  // sum_j m[j] s[j] (+ extra 2^256-shifted addend), in chunks of Wide::kWideTerms terms
  auto dot = [&](const Fr* m, const Fr* sv, int n) {
    Fr::Wide w;
    int used = 0;
    for (int j = 0; j < n; ++j) {
      if (used == Fr::Wide::kWideTerms - 1) {  // fold what is there and carry it on as one shifted term
        Fr part = w.reduce();
        w = Fr::Wide();
        w.add_shifted(part);
        used = 1;
      }
      w.add_product(m[j], sv[j]);
      ++used;
    }
    return w.reduce();
  };
  auto dense = [&](const std::vector<Fr>& m) {
    for (int i = 0; i < t; ++i) next[i] = dot(&m[(size_t)i * t], state, t);
    for (int i = 0; i < t; ++i) state[i] = next[i];
  };
  for (int i = 0; i < t; ++i) state[i] = state[i] + o.pre[(size_t)i];
  size_t fk = 0;
  for (int r = 0; r < h; ++r) {  // first half: full rounds, the last one with the pre-sparse matrix
    for (int i = 0; i < t; ++i) state[i] = sbox(state[i]) + o.full_k[fk][(size_t)i];
    ++fk;
    dense(r + 1 < h ? o.mds : o.pre_sparse);
  }
  for (int r = 0; r < o.r_p; ++r) {  // partial rounds: one S-box, sparse matrix
    state[0] = sbox(state[0]) + o.partial_k[(size_t)r];
    const Fr* row = o.sparse_row[(size_t)r].data();
    const Fr* col = o.sparse_col[(size_t)r].data();
    const Fr s0 = state[0];
    const Fr acc = dot(row, state, t);
    for (int i = 1; i < t; ++i) {  // state[i] + col[i - 1] s0: one product, the addend rides along as state[i] 2^256
      Fr::Wide w;
      w.add_shifted(state[i]);
      w.add_product(col[i - 1], s0);
      state[i] = w.reduce();
    }
    state[0] = acc;
  }
  for (int r = 0; r < h; ++r) {  // second half: full rounds, no constant after the last S-box
    const bool last = r + 1 == h;
    for (int i = 0; i < t; ++i) state[i] = last ? sbox(state[i]) : sbox(state[i]) + o.full_k[fk][(size_t)i];
    if (!last) ++fk;
    dense(o.mds);
  }
  for (int i = 0; i < t; ++i) state_v[(size_t)i] = state[i];
}

// the lane-layout tables of poseidon_ifma.hpp, from the optimised schedule's
inline const poseidon_ifma::Tables* poseidon_ifma_tables(int t, int r_f, int r_p) {
#if defined(__x86_64__) && defined(__GNUC__)
  namespace pi = poseidon_ifma;
  if (!pi::available() || t > 8) return nullptr;
  static std::map<std::tuple<int, int, int>, pi::Tables> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(t, r_f, r_p);
  auto it = cache.find(key);
  if (it != cache.end()) return &it->second;
  const PoseidonOpt& o = poseidon_opt(t, r_f, r_p);
  pi::Tables T;
  T.t = t, T.r_f = r_f, T.r_p = r_p;
  T.np = Fr::INV & pi::kMask52;  // -r^-1 mod 2^64, cut to 2^52
  uint64_t pl[5];
  pi::split52(Fr::MOD, pl);
  T.p = pi::zero(), T.one = pi::zero(), T.pre = pi::zero();
  for (int lane = 0; lane < 8; ++lane) {
    for (int k = 0; k < 5; ++k) T.p.l[k][lane] = pl[k];
    pi::set_lane(T.one, lane, Fr::one());
  }
  auto vec = [&](const std::vector<Fr>& src) {  // lane i = src[i]
    pi::V v = pi::zero();
    for (size_t i = 0; i < src.size() && i < 8; ++i) pi::set_lane(v, (int)i, src[i]);
    return v;
  };
  T.pre = vec(o.pre);
  for (auto& k : o.full_k) T.full_k.push_back(vec(k));
  auto cols = [&](const std::vector<Fr>& m) {
    std::vector<pi::V> out;
    for (int j = 0; j < t; ++j) {
      pi::V v = pi::zero();
      for (int i = 0; i < t; ++i) pi::set_lane(v, i, m[(size_t)i * t + j]);
      out.push_back(v);
    }
    return out;
  };
  T.mds_col = cols(o.mds);
  T.pre_sparse_col = cols(o.pre_sparse);
  for (int r = 0; r < r_p; ++r) {
    pi::V k = pi::zero();
    pi::set_lane(k, 0, o.partial_k[(size_t)r]);
    T.partial_k.push_back(k);
    T.row.push_back(vec(o.sparse_row[(size_t)r]));
    pi::V c = pi::zero();
    for (int i = 1; i < t; ++i) pi::set_lane(c, i, o.sparse_col[(size_t)r][(size_t)i - 1]);
    T.col.push_back(c);
    // the three-product form: c = (row_0, col_1 .. col_{t-1}), c k, the row without its first entry
    pi::V cv = c, ckv = pi::zero(), rr = T.row.back();
    const Fr& kr = o.partial_k[(size_t)r];
    pi::set_lane(cv, 0, o.sparse_row[(size_t)r][0]);
    pi::set_lane(ckv, 0, o.sparse_row[(size_t)r][0] * kr);
    for (int i = 1; i < t; ++i) pi::set_lane(ckv, i, o.sparse_col[(size_t)r][(size_t)i - 1] * kr);
    for (int k = 0; k < 5; ++k) rr.l[k][0] = 0;
    T.cvec.push_back(cv);
    T.ck.push_back(ckv);
    T.row_rest.push_back(rr);
  }
  return &cache.emplace(key, std::move(T)).first->second;
#else
  (void)t, (void)r_f, (void)r_p;
  return nullptr;
#endif
}

// state <- permutation(state) on the IFMA path (test hook and the sponge below); false if the CPU has no AVX-512 IFMA
inline bool poseidon_permute_ifma(std::vector<Fr>& state, int r_f, int r_p) {
#if defined(__x86_64__) && defined(__GNUC__)
  const poseidon_ifma::Tables* T = poseidon_ifma_tables((int)state.size(), r_f, r_p);
  if (!T) return false;
  poseidon_ifma::V v = poseidon_ifma::zero();
  for (size_t i = 0; i < state.size(); ++i) poseidon_ifma::set_lane(v, (int)i, state[i]);
  poseidon_ifma::permute(v, *T);
  for (size_t i = 0; i < state.size(); ++i) {
    uint64_t l[5];
    for (int k = 0; k < 5; ++k) l[k] = v.l[k][i];
    state[i] = poseidon_ifma::fr_from_limbs(l);
  }
  return true;
#else
  (void)state, (void)r_f, (void)r_p;
  return false;
#endif
}

// poseidon.rs:115-202: sponge framing over the permutation.  On a CPU with AVX-512 IFMA the state lives in the lane layout
// of poseidon_ifma.hpp for the life of the sponge (inputs converted on the way in, a challenge on the way out).
class Poseidon {
 public:
  Poseidon(int t, int rate, int r_f, int r_p)
      : t_(t), rate_(rate), spec_(&poseidon_spec(t, r_f, r_p)), ifma_(poseidon_ifma_tables(t, r_f, r_p)) {
    // poseidon::State::default(): first word 2^64
    Fr two32 = Fr::from_u64(1ull << 32);
    const Fr w0 = two32 * two32;
#if defined(__x86_64__) && defined(__GNUC__)
    if (ifma_) {
      vstate_ = poseidon_ifma::zero();
      poseidon_ifma::set_lane(vstate_, 0, w0);
      return;
    }
#endif
    state_.assign((size_t)t, Fr::zero());
    state_[0] = w0;
  }
  // :145-147.  The reference only buffers here and runs every permutation inside `squeeze`.  A complete chunk of `rate`
  // elements is absorbed the same way whenever it is absorbed (no padding word after a full-rate chunk, :61-74), so an
  // EAGER sponge -- `set_eager(true)`: permute as soon as a chunk is complete -- leaves `squeeze` the partial chunk (or the
  // empty permutation after an exact multiple: `buf_` is then empty, and `exact` below is decided by the same remainder)
  // and returns the same challenge.  It is what lets a long absorb overlap the work that produces its input
  // (aggregation.hpp: the accumulation transcript of a pipelined job).
  void set_eager(bool on) { eager_ = on; }
  void update(const std::vector<Fr>& elements) {
    buf_.insert(buf_.end(), elements.begin(), elements.end());
    if (eager_ && buf_.size() >= (size_t)rate_) {
      size_t i = 0;
      for (; i + (size_t)rate_ <= buf_.size(); i += (size_t)rate_) permutation(buf_.data() + i, (size_t)rate_);
      buf_.erase(buf_.begin(), buf_.begin() + (ptrdiff_t)i);
    }
  }
  Fr squeeze() {                                                                                               // :151-164
    std::vector<Fr> buf;
    buf.swap(buf_);
    const bool exact = buf.size() % (size_t)rate_ == 0;
    for (size_t i = 0; i < buf.size(); i += (size_t)rate_) {
      size_t n = std::min((size_t)rate_, buf.size() - i);
      permutation(buf.data() + i, n);
    }
    if (exact) permutation(nullptr, 0);
#if defined(__x86_64__) && defined(__GNUC__)
    if (ifma_) {
      uint64_t l[5];
      for (int k = 0; k < 5; ++k) l[k] = vstate_.l[k][1];
      return poseidon_ifma::fr_from_limbs(l);
    }
#endif
    return state_[1];
  }

 private:
  void permutation(const Fr* inputs, size_t n) {  // :44-75 (absorb + the 1 after the last input), :166-201
#if defined(__x86_64__) && defined(__GNUC__)
    if (ifma_) {
      poseidon_ifma::V add = poseidon_ifma::zero();
      for (size_t i = 0; i < n; ++i) poseidon_ifma::set_lane(add, 1 + (int)i, inputs[i]);
      if (1 + n < (size_t)t_) poseidon_ifma::set_lane(add, 1 + (int)n, Fr::one());
      poseidon_ifma::absorb(vstate_, add);
      poseidon_ifma::permute(vstate_, *ifma_);
      return;
    }
#endif
    for (size_t i = 0; i < n; ++i) state_[1 + i] = state_[1 + i] + inputs[i];
    if (1 + n < (size_t)t_) state_[1 + n] = state_[1 + n] + Fr::one();  // nothing after a full-rate chunk (:61-74)
    poseidon_permute(state_, *spec_);
  }
  int t_, rate_;
  const PoseidonSpec* spec_;
  const poseidon_ifma::Tables* ifma_;
#if defined(__x86_64__) && defined(__GNUC__)
  poseidon_ifma::V vstate_;
#endif
  std::vector<Fr> state_;
  std::vector<Fr> buf_;
  bool eager_ = false;
};

namespace fq_host {
inline void sub_mod(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {  // a, b < p
  unsigned __int128 br = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 x = (unsigned __int128)a[i] - b[i] - (uint64_t)br;
    t[i] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
  if (br) {
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (unsigned __int128)t[i] + P[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  memcpy(r, t, 32);
}
// sqrt for p = 3 (mod 4): a^((p+1)/4); false if a is not a square.  The ladder runs in the
// Montgomery domain (one product per step).
inline bool sqrt_mod(uint64_t r[4], const uint64_t a[4]) {
  // (p + 1) / 4
  static const uint64_t E[4] = {0x4f082305b61f3f52ull, 0x65e05aa45a1c72a3ull, 0x6e14116da0605617ull,
                                0x0c19139cb84c680aull};
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull,
                                     0x06d89f71cab8351full};
  const uint64_t one[4] = {1, 0, 0, 0};
  uint64_t am[4], acc[4];
  mont_mul(am, a, R2);     // a R
  mont_mul(acc, one, R2);  // 1 R
  for (int i = 252; i >= 0; --i) {  // E < 2^253
    mont_mul(acc, acc, acc);
    if ((E[i >> 6] >> (i & 63)) & 1) mont_mul(acc, acc, am);
  }
  uint64_t y[4], chk[4];
  mont_mul(y, acc, one);  // out of the Montgomery domain
  mul_mod(chk, y, y);
  if (memcmp(chk, a, 32) != 0) return false;
  memcpy(r, y, 32);
  return true;
}
}  // namespace fq_host
// (p + 1) / 4, the exponent of the square root (four 64-bit words, little-endian)
inline const uint64_t* fq_host_sqrt_exponent() {
  static const uint64_t E[4] = {0x4f082305b61f3f52ull, 0x65e05aa45a1c72a3ull, 0x6e14116da0605617ull, 0x0c19139cb84c680aull};
  return E;
}

// halo2curves 0.6.0 bn256 `G1Affine::{to_bytes, from_bytes}` (compressed, 32 bytes) as recalled
inline void g1_compress(const G1Affine& p, uint8_t out[32]) {
  if (p.is_identity()) {
    memset(out, 0, 32);
    out[31] |= 0x80;
    return;
  }
  memcpy(out, p.b, 32);
  out[31] |= (uint8_t)((p.b[32] & 1) << 6);
}
// returns false for an invalid encoding; the identity decodes to 64 zero bytes
inline bool g1_decompress(const uint8_t in[32], G1Affine* out) {
  uint8_t xb[32];
  memcpy(xb, in, 32);
  const int is_inf = xb[31] >> 7, ysign = (xb[31] >> 6) & 1;
  xb[31] &= 0x3F;
  uint64_t x[4];
  memcpy(x, xb, 32);
  if (!fq_host::lt_p(x)) return false;
  if (is_inf) {  // the identity has exactly one encoding: flag set, everything else zero
    if ((x[0] | x[1] | x[2] | x[3]) != 0 || ysign) return false;
    *out = G1Affine::identity();
    return true;
  }
  uint64_t x2[4], x3[4], three[4] = {3, 0, 0, 0}, y2[4], y[4];
  fq_host::mul_mod(x2, x, x);
  fq_host::mul_mod(x3, x2, x);
  fq_host::add_mod(y2, x3, three);
  if (!fq_host::sqrt_mod(y, y2)) return false;
  if ((int)(y[0] & 1) != ysign) {
    uint64_t zero[4] = {0, 0, 0, 0};
    fq_host::sub_mod(y, zero, y);
  }
  memcpy(out->b, x, 32);
  memcpy(out->b + 32, y, 32);
  return true;
}

// Up to 8 compressed points decoded TOGETHER on AVX-512 IFMA: the square roots -- a^((p+1)/4), 253 squarings and 64 products
// with a 4-bit window, a dependency chain of ~6.5 us -- run one per lane (poseidon_ifma.hpp's lane-wise Montgomery
// product, modulus p, R = 2^260), so a group of points costs what ONE costs in scalar code (9.5 us each).  A transcript
// uses this for the points it is about to read in one go (`read_n_ec_points`: a proof's witness / quotient / opening
// commitments), as HINTS: read_ec_point takes a decoding only if it is the decoding of the bytes (`hint_matches`), and
// identities, invalid encodings and non-residues are left to `g1_decompress` -- verdicts and error texts are its own.
// ok[i] = 1: out[i] is a finite curve point whose x / parity are the encoded ones.  False if the CPU has no IFMA.
inline bool g1_decompress_x8(const uint8_t* const enc[8], size_t n, G1Affine out[8], uint8_t ok[8]) {
#if defined(__x86_64__) && defined(__GNUC__)
  namespace pi = poseidon_ifma;
  if (!pi::available() || n == 0 || n > 8) return false;
  struct Consts {
    pi::V p, r2, one_m, three_m, one_plain;
    uint64_t np;
  };
  static const Consts C = [] {
    Consts c;
    uint64_t pl[5];
    pi::split52(fq_host::P, pl);
    // -p^-1 mod 2^52 by Newton's iteration on the low limb
    uint64_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - pl[0] * inv;
    c.np = (0 - inv) & pi::kMask52;
    auto pow2 = [](int e, uint64_t out4[4]) {  // 2^e mod p by doublings
      uint64_t v[4] = {1, 0, 0, 0};
      for (int i = 0; i < e; ++i) fq_host::add_mod(v, v, v);
      memcpy(out4, v, 32);
    };
    uint64_t r1[4], r2[4], three[4];
    pow2(260, r1);
    pow2(520, r2);
    fq_host::add_mod(three, r1, r1);
    fq_host::add_mod(three, three, r1);
    uint64_t l[5], lr2[5], l3[5];
    pi::split52(r1, l);
    pi::split52(r2, lr2);
    pi::split52(three, l3);
    for (int lane = 0; lane < 8; ++lane)
      for (int k = 0; k < 5; ++k) {
        c.p.l[k][lane] = pl[k];
        c.r2.l[k][lane] = lr2[k];
        c.one_m.l[k][lane] = l[k];
        c.three_m.l[k][lane] = l3[k];
        c.one_plain.l[k][lane] = k == 0 ? 1 : 0;
      }
    return c;
  }();
  uint64_t xs[8][4];
  int ysign[8];
  pi::V xv = pi::zero();
  for (size_t i = 0; i < 8; ++i) {
    ok[i] = 0;
    if (i >= n) continue;
    uint8_t xb[32];
    memcpy(xb, enc[i], 32);
    const int is_inf = xb[31] >> 7;
    ysign[i] = (xb[31] >> 6) & 1;
    xb[31] &= 0x3F;
    memcpy(xs[i], xb, 32);
    if (is_inf || !fq_host::lt_p(xs[i])) continue;  // (left to g1_decompress)
    ok[i] = 1;
    uint64_t l[5];
    pi::split52(xs[i], l);
    for (int k = 0; k < 5; ++k) xv.l[k][i] = l[k];
  }
  pi::V yv;
  pi::sqrt_x3_plus_b(xv, C.p, C.np, C.r2, C.one_m, C.three_m, C.one_plain, fq_host_sqrt_exponent(), yv);
  for (size_t i = 0; i < n; ++i) {
    if (!ok[i]) continue;
    uint64_t l[5], y[4];
    for (int k = 0; k < 5; ++k) l[k] = yv.l[k][i];
    pi::join52(l, y);  // below 4 p: canonical by at most three subtractions
    for (int k = 0; k < 3 && !fq_host::lt_p(y); ++k) {
      unsigned __int128 br = 0;
      for (int w = 0; w < 4; ++w) {
        unsigned __int128 d = (unsigned __int128)y[w] - fq_host::P[w] - (uint64_t)br;
        y[w] = (uint64_t)d;
        br = (d >> 64) & 1;
      }
    }
    // y^2 = x^3 + 3 in scalar arithmetic: a non-residue (or any slip of the vector path) leaves the point to the scalar decoder
    uint64_t x2[4], x3[4], rhs[4], three[4] = {3, 0, 0, 0}, chk[4];
    fq_host::mul_mod(x2, xs[i], xs[i]);
    fq_host::mul_mod(x3, x2, xs[i]);
    fq_host::add_mod(rhs, x3, three);
    fq_host::mul_mod(chk, y, y);
    if (!fq_host::lt_p(y) || memcmp(chk, rhs, 32) != 0) {
      ok[i] = 0;
      continue;
    }
    if ((int)(y[0] & 1) != ysign[i]) {
      uint64_t zero[4] = {0, 0, 0, 0};
      fq_host::sub_mod(y, zero, y);
    }
    memcpy(out[i].b, xs[i], 32);
    memcpy(out[i].b + 32, y, 32);
  }
  return true;
#else
  (void)enc, (void)n, (void)out, (void)ok;
  return false;
#endif
}

// Sponge policies for the transcript below.  `Poseidon` (above) hashes on the host.  The other
// two split a transcript into "what was absorbed, where were the squeezes" and "parse again with
// the challenges known": the sequence of absorb / squeeze operations of a proof is fixed by the
// protocol, so the hashing of MANY proofs can be one device launch
// (`snarkv_poseidon_transcript_batch`, csrc/poseidon.hip) between two cheap parsing passes.
struct RecordingSponge {
  std::vector<Fr> elems;          // every absorbed element, in order
  std::vector<uint32_t> seg_len;  // elements absorbed before each squeeze
  uint32_t cur = 0;
  RecordingSponge(int, int, int, int) {
    elems.reserve(64);
    seg_len.reserve(8);
  }
  void update(const std::vector<Fr>& e) {
    elems.insert(elems.end(), e.begin(), e.end());
    cur += (uint32_t)e.size();
  }
  Fr squeeze() {
    seg_len.push_back(cur);
    cur = 0;
    return Fr::one();  // placeholder: reading a proof only STORES challenges
  }
};
struct ReplaySponge {
  std::vector<Fr> challenges;
  size_t next = 0;
  ReplaySponge(int, int, int, int) {}
  void update(const std::vector<Fr>&) {}
  Fr squeeze() {
    if (next >= challenges.size()) throw Panic("ReplaySponge: more squeezes than recorded challenges");
    return challenges[next++];
  }
};

template <class Sponge>
class PoseidonTranscriptT : public Transcript {
 public:
  // T = 5, RATE = 4, R_F = 8, R_P = 60: examples/evm-verifier-with-accumulator.rs:36-39
  explicit PoseidonTranscriptT(std::vector<uint8_t> proof = {}, int t = 5, int rate = 4, int r_f = 8, int r_p = 60)
      : stream_(std::move(proof)), buf_(t, rate, r_f, r_p) {}
  Sponge& sponge() { return buf_; }

  Fr squeeze_challenge() override { return buf_.squeeze(); }  // halo2.rs:211-213
  Error common_scalar(const Fr& s) override {                 // halo2.rs:215-218
    if (record_layout_) {
      if (pending_src_ != kNoSrc) {
        layout_.push_back(pending_src_);
      } else {
        layout_.push_back((kSrcLead << 28) | (uint32_t)lead_.size());
        lead_.push_back(s);
      }
      pending_src_ = kNoSrc;
    }
    buf_.update({s});
    return Error{};
  }
  // halo2.rs:220-237: x and y go through `fe_to_fe` (Fq -> integer -> mod r)
  Error common_ec_point(const G1Affine& p) override {
    if (p.is_identity()) return Error{Error::Transcript, "Invalid elliptic curve point encoding in proof"};
    uint64_t w[4];
    memcpy(w, p.b, 32);
    Fr x = grain::fr_from_words_mod_r(w);  // p < 2r: one conditional subtraction
    memcpy(w, p.b + 32, 32);
    Fr y = grain::fr_from_words_mod_r(w);
    if (record_layout_) {
      if (pending_src_ != kNoSrc) {  // point number q of the proof: its x, then its y
        layout_.push_back((kSrcPx << 28) | pending_src_);
        layout_.push_back((kSrcPy << 28) | pending_src_);
      } else {
        layout_.push_back((kSrcLead << 28) | (uint32_t)lead_.size());
        layout_.push_back((kSrcLead << 28) | (uint32_t)(lead_.size() + 1));
        lead_.push_back(x);
        lead_.push_back(y);
      }
      pending_src_ = kNoSrc;
    }
    buf_.update({x, y});
    return Error{};
  }
  Result<Fr> read_scalar() override {  // halo2.rs:247-260: 32 bytes little-endian canonical
    if (pos_ + 32 > stream_.size()) return Result<Fr>::Err(Error{Error::Transcript, "failed to fill whole buffer"});
    Fr s;
    bool ok = Fr::from_bytes(stream_.data() + pos_, &s);
    pos_ += 32;
    if (!ok) return Result<Fr>::Err(Error{Error::Transcript, "Invalid scalar encoding in proof"});
    if (record_layout_) pending_src_ = (kSrcScalar << 28) | (uint32_t)(pos_ - 32);
    common_scalar(s);
    return Result<Fr>::Ok(s);
  }
  // The next n points sit side by side in the stream: decode (up to 8 of) them together -- one square-root chain for the
  // group instead of one per point.  Skipped when the decodings come from elsewhere (a replay pass, device hints).
  void prefetch_points(size_t n) override {
    pf_n_ = 0;
    const size_t first = point_offsets_.size();
    if (next_decoded_ < n_decoded_in_ || first < n_hints_ || getenv("SNARKV_HOST_NO_POINT_PREFETCH")) return;
    n = std::min<size_t>(std::min<size_t>(n, kPrefetchMax), (stream_.size() - std::min(pos_, stream_.size())) / 32);
    if (n < 2) return;
    for (size_t g = 0; g < n; g += 8) {  // groups of eight lanes
      const size_t m = std::min<size_t>(8, n - g);
      const uint8_t* enc[8];
      for (size_t i = 0; i < m; ++i) enc[i] = stream_.data() + pos_ + 32 * (g + i);
      if (!g1_decompress_x8(enc, m, pf_pts_ + g, pf_ok_ + g)) return;  // (no IFMA: pf_n_ stays 0)
    }
    pf_pos_ = pos_, pf_n_ = n;
  }
  Result<G1Affine> read_ec_point() override {  // halo2.rs:262-275: compressed `C::from_bytes`
    if (pos_ + 32 > stream_.size())
      return Result<G1Affine>::Err(Error{Error::Transcript, "failed to fill whole buffer"});
    G1Affine p;
    bool ok = true;
    const uint8_t* enc = stream_.data() + pos_;
    const size_t read_index = point_offsets_.size();
    point_offsets_.push_back(pos_);
    if (next_decoded_ < n_decoded_in_) {
      p = decoded_in_[next_decoded_++];  // second parsing pass: the square root was taken in the first
    } else if (read_index < n_hints_ && hint_ok_[read_index] && hint_matches(hint_pts_ + 64 * read_index, enc)) {
      memcpy(p.b, hint_pts_ + 64 * read_index, 64);  // decompressed by the device for the whole batch (snarkv_g1_decompress)
    } else if (pf_n_ && pos_ >= pf_pos_ && (pos_ - pf_pos_) % 32 == 0 && (pos_ - pf_pos_) / 32 < pf_n_ &&
               pf_ok_[(pos_ - pf_pos_) / 32] && hint_matches(pf_pts_[(pos_ - pf_pos_) / 32].b, enc)) {
      p = pf_pts_[(pos_ - pf_pos_) / 32];  // decoded with its neighbours by `prefetch_points` (g1_decompress_x8)
    } else {
      ok = g1_decompress(enc, &p);
      // Fused device route: the challenges were hashed over the DEVICE's decoding of these bytes.  A finite point the host
      // decodes where the device's answer was unusable means the two disagree -- the proof would be checked under
      // challenges that are not the hash of what is parsed here.  Refuse it (defence in depth: unreachable while the two
      // decoders agree on every encoding, tests/test_gpu_poseidon.py).
      if (ok && strict_hints_ && read_index < n_hints_ && !p.is_identity())
        return Result<G1Affine>::Err(Error{Error::Transcript, "device and host disagree on a compressed point of the proof"});
    }
    pos_ += 32;
    if (!ok) return Result<G1Affine>::Err(Error{Error::Transcript, "Invalid elliptic curve point encoding in proof"});
    decoded_.push_back(p);
    if (record_layout_) pending_src_ = (uint32_t)read_index;
    Error e = common_ec_point(p);  // the identity decodes but has no coordinates to absorb
    pending_src_ = kNoSrc;
    if (!e.ok()) return Result<G1Affine>::Err(e);
    return Result<G1Affine>::Ok(p);
  }
  Error write_scalar(const Fr& s) override {  // halo2.rs:300-309
    common_scalar(s);
    uint8_t le[32];
    s.to_bytes(le);
    stream_.insert(stream_.end(), le, le + 32);
    return Error{};
  }
  Error write_ec_point(const G1Affine& p) override {  // halo2.rs:311-320
    Error e = common_ec_point(p);
    if (!e.ok()) return e;
    uint8_t c[32];
    g1_compress(p, c);
    stream_.insert(stream_.end(), c, c + 32);
    return Error{};
  }
  const std::vector<uint8_t>& stream() const { return stream_; }
  std::vector<uint8_t> finalize() { return std::move(stream_); }
  size_t remaining() const { return stream_.size() - pos_; }

 private:
  std::vector<uint8_t> stream_;
  size_t pos_ = 0;
  Sponge buf_;
  std::vector<G1Affine> decoded_;               // points decompressed by this pass ...
  const G1Affine* decoded_in_ = nullptr;        // ... / handed over by an earlier one (borrowed: the caller keeps them alive)
  size_t n_decoded_in_ = 0, next_decoded_ = 0;
  std::vector<size_t> point_offsets_;           // stream position of every point read so far
  const uint8_t* hint_pts_ = nullptr;           // candidate decodings by read index (64 bytes each) and their validity
  const uint8_t* hint_ok_ = nullptr;            // flags: borrowed views of the device's answer for the whole batch --
  size_t n_hints_ = 0;                          // no per-proof copies, nothing for a pool worker to free
  // the points of the current `read_n_ec_points`, decoded together (prefetch_points): bytes [pf_pos_, pf_pos_ + 32 pf_n_)
  static constexpr size_t kPrefetchMax = 32;
  G1Affine pf_pts_[kPrefetchMax];
  uint8_t pf_ok_[kPrefetchMax] = {};
  size_t pf_pos_ = 0, pf_n_ = 0;
  bool strict_hints_ = false;
  bool record_layout_ = false;
  uint32_t pending_src_ = 0xFFFFFFFFu;          // the source of the element(s) the next common_* call absorbs
  std::vector<uint32_t> layout_;
  std::vector<Fr> lead_;
  // A hint is used only if it IS the decoding of these 32 bytes: a finite point whose x equals the encoded x and whose
  // y has the encoded parity (on the curve by construction: the device checks y^2 = x^3 + 3 before answering).
  // Identities, invalid encodings and anything that does not match go through g1_decompress as before.
  static bool hint_matches(const uint8_t h[64], const uint8_t enc[32]) {
    if (enc[31] >> 7) return false;
    bool zero = true;
    for (int i = 0; i < 64 && zero; ++i) zero = h[i] == 0;
    if (zero) return false;
    return memcmp(enc, h, 31) == 0 && (enc[31] & 0x3F) == h[31] && ((enc[31] >> 6) & 1) == (h[32] & 1);
  }

 public:
  std::vector<G1Affine>& decoded_points() { return decoded_; }
  void set_decoded_points(const G1Affine* pts, size_t n) {
    decoded_in_ = pts, n_decoded_in_ = n, next_decoded_ = 0;
  }
  // where this pass read its points (the layout is the protocol's: the same for every proof of it)
  const std::vector<size_t>& point_offsets() const { return point_offsets_; }
  // Where every absorbed element came from, in the codes of `snarkv_poseidon_read_batch` (include/snarkv_amd.h):
  // kind << 28 | value, kind 0 = a value the caller brought (`lead_values()[value]`: initial state, instances),
  // 1 = the proof's scalar at byte `value`, 2 / 3 = x / y of the proof's point number `value`.
  static constexpr uint32_t kSrcLead = 0, kSrcScalar = 1, kSrcPx = 2, kSrcPy = 3, kNoSrc = 0xFFFFFFFFu;
  void record_layout() { record_layout_ = true; }
  const std::vector<uint32_t>& layout() const { return layout_; }
  const std::vector<Fr>& lead_values() const { return lead_; }
  // decodings computed elsewhere for the k-th point read, checked against the bytes before use
  // `strict`: the caller's challenges depend on these decodings (the fused device route): see read_ec_point
  void set_point_hints(const uint8_t* pts64, const uint8_t* ok, size_t n, bool strict = false) {
    hint_pts_ = pts64, hint_ok_ = ok, n_hints_ = n, strict_hints_ = strict;
  }
};
using PoseidonTranscript = PoseidonTranscriptT<Poseidon>;

// The optimised tables as 32-byte LE rows, in the order `snarkv_poseidon_create` takes them
// (= the fields of the reference's `poseidon::Spec`).
struct PoseidonTableBytes {
  std::vector<uint8_t> start, partial, end, mds, pre_sparse, rows, cols;
};
inline PoseidonTableBytes poseidon_table_bytes(int t, int r_f, int r_p) {
  const PoseidonOpt& o = poseidon_opt(t, r_f, r_p);
  PoseidonTableBytes b;
  auto put = [](std::vector<uint8_t>& v, const Fr& x) {
    size_t k = v.size();
    v.resize(k + 32);
    x.to_bytes(&v[k]);
  };
  const int h = r_f / 2;
  for (auto& x : o.pre) put(b.start, x);
  for (int r = 0; r < h; ++r)
    for (auto& x : o.full_k[(size_t)r]) put(b.start, x);
  for (auto& x : o.partial_k) put(b.partial, x);
  for (size_t r = (size_t)h; r < o.full_k.size(); ++r)
    for (auto& x : o.full_k[r]) put(b.end, x);
  for (auto& x : o.mds) put(b.mds, x);
  for (auto& x : o.pre_sparse) put(b.pre_sparse, x);
  for (auto& row : o.sparse_row)
    for (auto& x : row) put(b.rows, x);
  for (auto& col : o.sparse_col)
    for (auto& x : col) put(b.cols, x);
  if (b.end.empty()) b.end.resize(32);  // r_f = 2: no rows, but a valid pointer
  return b;
}

}  // namespace snarkv_host
