// BN254 scalar field Fr on the HOST: the `LoadedScalar` of the native loader.
//
// The reference keeps every Fr computation of the accumulation path on the
// host CPU (powers of r/u/v/mu/gamma, barycentric weights, batch inversion:
// snark-verifier/src/pcs/kzg/accumulation.rs:52, multiopen/gwc19.rs:52-76,
// multiopen/bdfg21.rs:173-223) and only the EC work reaches the loader's
// `multi_scalar_multiplication`.  This mirror does the same: Fr here, G1/pairing
// on the MI355X.  4 x 64-bit Montgomery (R = 2^256), the halo2curves layout.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

namespace snarkv_host {

class Fr {
 public:
  uint64_t v[4];  // Montgomery form

#if defined(SNARKV_HOST_PALLAS)
  // the pasta flavour of the host mirror (host/test_driver_pallas.cpp): `pallas::Scalar`,
  // q = 2^254 + 45560315531506369815346746415080538113
  static constexpr uint64_t MOD[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0000000000000000ull,
                                      0x4000000000000000ull};
  static constexpr uint64_t INV = 0x8c46eb20ffffffffull;  // -q^-1 mod 2^64
  static constexpr uint64_t ONE_M[4] = {0x5b2b3e9cfffffffdull, 0x992c350be3420567ull, 0xffffffffffffffffull,
                                        0x3fffffffffffffffull};
  static constexpr uint64_t R2[4] = {0xfc9678ff0000000full, 0x67bb433d891a16e3ull, 0x7fae231004ccf590ull,
                                     0x096d41af7ccfdaa9ull};
#else
  static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull,
                                      0x30644e72e131a029ull};
  static constexpr uint64_t INV = 0xc2e1f593efffffffull;  // -r^-1 mod 2^64
  static constexpr uint64_t ONE_M[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull,
                                        0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull,
                                     0x0216d0b17f4e44a5ull};

#endif

  Fr() : v{0, 0, 0, 0} {}
  static Fr zero() { return Fr(); }
  static Fr one() {
    Fr r;
    memcpy(r.v, ONE_M, 32);
    return r;
  }
  static Fr from_u64(uint64_t x) {
    Fr t;
    t.v[0] = x;
    return mont_mul(t, r2());
  }
  // 32-byte little-endian canonical (`PrimeField::from_repr`); false if >= r
  static bool from_bytes(const uint8_t b[32], Fr* out) {
    Fr t;
    memcpy(t.v, b, 32);
    if (!lt_mod(t.v)) return false;
    *out = mont_mul(t, r2());
    return true;
  }
  void to_bytes(uint8_t b[32]) const {  // `PrimeField::to_repr`
    Fr one_raw;
    one_raw.v[0] = 1;
    Fr t = mont_mul(*this, one_raw);
    memcpy(b, t.v, 32);
  }
  bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
  bool operator==(const Fr& o) const { return memcmp(v, o.v, 32) == 0; }
  bool operator!=(const Fr& o) const { return !(*this == o); }

  Fr operator+(const Fr& o) const {
    Fr r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (unsigned __int128)v[i] + o.v[i];
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
    reduce_once(r.v);
    return r;
  }
  Fr operator-(const Fr& o) const {
    Fr r;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 x = (unsigned __int128)v[i] - o.v[i] - (uint64_t)br;
      r.v[i] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
    if (br) {
      unsigned __int128 c = 0;
      for (int i = 0; i < 4; ++i) {
        c += (unsigned __int128)r.v[i] + MOD[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  Fr operator-() const { return Fr() - *this; }
  Fr operator*(const Fr& o) const { return mont_mul(*this, o); }
  Fr& operator+=(const Fr& o) { return *this = *this + o; }
  Fr& operator-=(const Fr& o) { return *this = *this - o; }
  Fr& operator*=(const Fr& o) { return *this = *this * o; }
  Fr square() const { return mont_mul(*this, *this); }

  Fr pow(const uint64_t e[4]) const {
    Fr res = one();
    for (int i = 3; i >= 0; --i)
      for (int b = 63; b >= 0; --b) {
        res = res.square();
        if ((e[i] >> b) & 1) res = res * *this;
      }
    return res;
  }
  // `Field::invert` by Fermat: x^(r - 2).  ~380 products; kept as the cross-check of `invert` (tests/hosttest)
  bool invert_fermat(Fr* out) const {
    if (is_zero()) return false;
    uint64_t e[4] = {MOD[0] - 2, MOD[1], MOD[2], MOD[3]};
    *out = pow(e);
    return true;
  }
  // `Field::invert`: None (false) for zero.  Kaliski's almost-inverse (binary extended Euclid on the 256-bit
  // integers, shifts batched by count-trailing-zeros) followed by two Montgomery products that remove the 2^k it
  // leaves: ~3x faster than the exponentiation -- every proof's barycentric weights pay one inversion
  // (`L::batch_invert`, plonk.rs:66-70), a third of its Fr algebra before this.
  //   invariants   x s = sigma v 2^k ,  x r = -sigma u 2^k  (mod r),   u odd, gcd(u, v) = 1
  bool invert(Fr* out) const {
    if (is_zero()) return false;
    uint64_t u[4], w[4], r[4] = {0, 0, 0, 0}, s[4] = {1, 0, 0, 0};
    memcpy(u, MOD, 32);
    memcpy(w, v, 32);  // the Montgomery residue x = a R as an integer in [1, r)
    unsigned k = 0;
    bool neg = false;  // sigma = -1
    auto shr = [](uint64_t* a, unsigned n) {  // 1 <= n <= 63
      a[0] = (a[0] >> n) | (a[1] << (64 - n));
      a[1] = (a[1] >> n) | (a[2] << (64 - n));
      a[2] = (a[2] >> n) | (a[3] << (64 - n));
      a[3] >>= n;
    };
    auto shl = [](uint64_t* a, unsigned n) {
      a[3] = (a[3] << n) | (a[2] >> (64 - n));
      a[2] = (a[2] << n) | (a[1] >> (64 - n));
      a[1] = (a[1] << n) | (a[0] >> (64 - n));
      a[0] <<= n;
    };
    auto sub = [](uint64_t* a, const uint64_t* b) {  // a -= b, a >= b
      unsigned __int128 br = 0;
      for (int i = 0; i < 4; ++i) {
        unsigned __int128 x = (unsigned __int128)a[i] - b[i] - (uint64_t)br;
        a[i] = (uint64_t)x;
        br = (x >> 64) & 1;
      }
    };
    auto add = [](uint64_t* a, const uint64_t* b) {
      unsigned __int128 c = 0;
      for (int i = 0; i < 4; ++i) {
        c += (unsigned __int128)a[i] + b[i];
        a[i] = (uint64_t)c;
        c >>= 64;
      }
    };
    auto less = [](const uint64_t* a, const uint64_t* b) {
      for (int i = 3; i >= 0; --i)
        if (a[i] != b[i]) return a[i] < b[i];
      return false;
    };
    for (;;) {
      // w even (or the difference just taken): halve it, double r
      while ((w[0] & 1) == 0) {
        if ((w[0] | w[1] | w[2] | w[3]) == 0) goto done;
        unsigned tz = w[0] ? (unsigned)__builtin_ctzll(w[0]) : 63u;
        if (tz > 63) tz = 63;
        shr(w, tz);
        shl(r, tz);
        k += tz;
      }
      if (less(w, u)) {  // keep w >= u: swapping the pairs flips sigma
        for (int i = 0; i < 4; ++i) {
          uint64_t t = u[i];
          u[i] = w[i];
          w[i] = t;
          t = r[i];
          r[i] = s[i];
          s[i] = t;
        }
        neg = !neg;
      }
      sub(w, u);
      add(s, r);
    }
  done:
    // Kaliski's last step (v - u = 0) also doubles r once: u = 1 now and x r' = -sigma 2^(k+1) with r' = 2 r
    shl(r, 1);
    k += 1;
    while (!lt_mod(r)) sub(r, MOD);  // r < 4 r-modulus here
    Fr t;
    memcpy(t.v, r, 32);
    if (!neg) t = Fr() - t;  // x^-1 = -sigma r 2^-k
    // x^-1 2^k  ->  a^-1 R = x^-1 R^2 :  times R^2 2^-k, as Montgomery products by R^2 and by a power of two
    Fr p2;
    if (k >= 257) {
      t = mont_mul(t, r2());  // x^-1 2^k R
      unsigned e = 512 - k;   // <= 255
      p2.v[e >> 6] = 1ull << (e & 63);
      *out = mont_mul(t, p2);  // x^-1 2^k R 2^(512-k) / R = x^-1 R^2 ... / 2^-0
    } else {
      t = mont_mul(mont_mul(t, r2()), r2());  // x^-1 2^k R^2
      unsigned e = 256 - k;                   // 0..2
      p2.v[0] = 1ull << e;
      *out = mont_mul(t, p2);
    }
    return true;
  }
  // `LoadedScalar::powers` (reference loader.rs:71-78): 1, x, ..., x^(n-1)
  std::vector<Fr> powers(size_t n) const {
    std::vector<Fr> out;
    out.reserve(n);
    Fr cur = one();
    for (size_t i = 0; i < n; ++i) {
      out.push_back(cur);
      cur = cur * *this;
    }
    return out;
  }
  // Lazy dot products for chains of multiply-adds whose terms are known together (the Poseidon rounds of the accumulation
  // transcript: one sponge over 4 m elements, m + 1 dependent permutations on ONE thread): the 512-bit products are
  // summed as integers and reduced ONCE -- sum_j a_j b_j costs 16 limb products per term + 16 for the reduction instead
  // of 32 per term.  `add_shifted(x)` adds x 2^256, i.e. the Montgomery residue x itself after the reduction.
  // At most kWideTerms terms per accumulator (the sum stays below 2^512 for both curves' moduli).
  struct Wide {
    static constexpr int kWideTerms = 6;
    uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    inline void add_product(const Fr& a, const Fr& b) {
      // the 512-bit product in locals (registers), then ONE carry chain into the accumulator
      uint64_t p[8];
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; ++j) {
        c += (unsigned __int128)a.v[0] * b.v[j];
        p[j] = (uint64_t)c;
        c >>= 64;
      }
      p[4] = (uint64_t)c;
      for (int i = 1; i < 4; ++i) {
        c = 0;
        for (int j = 0; j < 4; ++j) {
          c += (unsigned __int128)a.v[i] * b.v[j] + p[i + j];
          p[i + j] = (uint64_t)c;
          c >>= 64;
        }
        p[i + 4] = (uint64_t)c;
      }
      c = 0;
      for (int k = 0; k < 8; ++k) {
        c += (unsigned __int128)w[k] + p[k];
        w[k] = (uint64_t)c;
        c >>= 64;
      }
    }
    inline void add_shifted(const Fr& x) {
      unsigned __int128 c = 0;
      for (int k = 0; k < 4; ++k) {
        c += (unsigned __int128)w[4 + k] + x.v[k];
        w[4 + k] = (uint64_t)c;
        c >>= 64;
      }
    }
    // (sum) / 2^256 mod r, fully reduced
    inline Fr reduce() const {
      uint64_t t[9];
      memcpy(t, w, 64);
      uint64_t carry = 0;  // the carry out of position i + 4, due at position i + 5
      for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[i] * INV;
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; ++j) {
          c += (unsigned __int128)m * MOD[j] + t[i + j];
          t[i + j] = (uint64_t)c;
          c >>= 64;
        }
        c += (unsigned __int128)t[i + 4] + carry;
        t[i + 4] = (uint64_t)c;
        carry = (uint64_t)(c >> 64);
      }
      t[8] = carry;
      Fr r;
      memcpy(r.v, t + 4, 32);
      uint64_t hi = t[8];
      while (hi || !lt_mod(r.v)) {  // < 4 r by the term bound: a few subtractions at most
        unsigned __int128 br = 0;
        for (int i = 0; i < 4; ++i) {
          unsigned __int128 x = (unsigned __int128)r.v[i] - MOD[i] - (uint64_t)br;
          r.v[i] = (uint64_t)x;
          br = (x >> 64) & 1;
        }
        hi -= (uint64_t)br;
      }
      return r;
    }
  };

  // total order on canonical values (the reference needs `Ord` for BTreeSet/Map keys)
  bool operator<(const Fr& o) const {
    uint8_t a[32], b[32];
    to_bytes(a);
    o.to_bytes(b);
    for (int i = 31; i >= 0; --i)
      if (a[i] != b[i]) return a[i] < b[i];
    return false;
  }

 private:
  static Fr r2() {
    Fr r;
    memcpy(r.v, R2, 32);
    return r;
  }
  static bool lt_mod(const uint64_t t[4]) {
    for (int i = 3; i >= 0; --i)
      if (t[i] != MOD[i]) return t[i] < MOD[i];
    return false;
  }
  static void reduce_once(uint64_t t[4]) {
    if (lt_mod(t)) return;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 x = (unsigned __int128)t[i] - MOD[i] - (uint64_t)br;
      t[i] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
  }
  static Fr mont_mul(const Fr& a, const Fr& b) {
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
      c = ((unsigned __int128)m * MOD[0] + t[0]) >> 64;
      for (int j = 1; j < 4; ++j) {
        c += (unsigned __int128)m * MOD[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r;
    memcpy(r.v, t, 32);
    reduce_once(r.v);
    return r;
  }
};

}  // namespace snarkv_host
