// The Poseidon permutation on AVX-512 IFMA (vpmadd52luq / vpmadd52huq): the HOST side of the transcripts.
//
// A transcript is a sponge -- a chain of dependent permutations on one thread -- and the accumulation transcript of an
// aggregation (`As::create_proof` over m accumulators, examples/evm-verifier-with-accumulator.rs:375) is ONE sponge over
// 4 m field elements: m + 1 permutations back to back, 1 025 of them at m = 1 024.  Neither more host threads nor the GPU
// shorten that chain (one permutation is ~270 DEPENDENT field products: a lone GPU lane needs ~150 us for it); what does
// is the width of one core.  The state (t <= 8 words) lives across the 8 lanes of zmm registers, one 260-bit value per
// lane in five 52-bit limbs, Montgomery with R = 2^260:
//   S-box          three lane-wise products (all words at once in a full round)
//   dense matrix   sum_j column_j * broadcast(state_j): t lane-wise products summed as 64-bit column accumulators, ONE
//                  Montgomery reduction for the t x t products of a round
//   sparse matrix  (partial rounds) row * state summed across the lanes + column * broadcast(s0) + state 2^260: two
//                  lane-wise products, one reduction -- folded with the S-box into THREE dependent products per round
//                  (`permute`: c x^5 = (c x) x^4 with c = (row_0, col_1 ..); the row's products share the fourth power's)
//   constants      ride along as value * 2^260 in the accumulators of the product that precedes them
// Operands of vpmadd52 must be below 2^52 per limb: every reduction ends in a carry pass.  Value bounds (r < 2^254,
// R = 2^260: a product of two values below 2^260 / 2^3 reduces to below 1.3 r without a conditional subtraction):
//   * AT PERMUTATION BOUNDARIES (what `fr_from_limbs` / `join52` may be given): below 4 r < 2^256;
//   * INSIDE the partial rounds words 1 .. t-1 are NOT multiplied: they gain up to + r per round (four-product form:
//     brought back every eighth round, ~9.5 r) or + 2 r per round (three-product form, the default: the round constant
//     rides along multiplied through; brought back every SIXTEENTH round, < 34 r) -- above 2^256, below 2^260 = 64.9 r:
//     the limbs stay below 2^52 and every vpmadd52 operand is legal, but such a mid-permutation state must never
//     reach `join52`.  Lengthening either period needs this bound re-derived (`permute` has the
//     derivation for the default form).
//   The worst case (all-(r-1) states, both parameter sets) is pinned against the scalar schedule in
//   tests/test_transcript.py::test_ifma_permutation_equals_the_scalar_schedule_and_the_oracle.
// The canonical residue is taken once, when a challenge leaves the sponge.  Same values as `poseidon_permute` (the scalar
// schedule it mirrors), word for word: tests/hosttest + tests/test_transcript.py pin one against the other.
// Where the time goes (EPYC 9575F, Zen 5): a dependent lane-wise product is 19.2 ns = 96 cycles (`mul_chain`), the
// permutation's chain is 60 x 3 + 8 x 4 = 212 of them = 4.1 us, measured 5.1 us.  The rest is the chain's own glue
// (three cross-lane broadcasts and two blends per partial round, the dense rounds' five accumulated products) -- NOT
// issue: folding the row's products into the fourth power's (365 -> 315 fused multiply-adds per round) moved the
// permutation from 5.17 to 5.06 us only.
// Selected at run time (`available()`): the library is built without -mavx512*, these functions carry their own target.
#pragma once
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>

#include "fr.hpp"

#define SNARKV_IFMA __attribute__((target("avx512f,avx512ifma,avx512dq,avx512bw,avx512vl"), always_inline)) inline
#define SNARKV_IFMA_FN __attribute__((target("avx512f,avx512ifma,avx512dq,avx512bw,avx512vl")))

namespace snarkv_host {
namespace poseidon_ifma {

inline bool available() {
  static const bool ok = [] {
    if (getenv("SNARKV_HOST_NO_IFMA")) return false;  // A/B and test knob: the scalar schedule
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq") &&
           __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
  }();
  return ok;
}

constexpr uint64_t kMask52 = (1ull << 52) - 1;

// A/B and test knob: 3 = the three-product partial round (default), 4 = the four-product form it replaced (the S-box's
// three products, then the row's): same values word for word (tests/test_transcript.py runs both)
inline int& partial_round_form() {
  static int form = [] {
    const char* e = getenv("SNARKV_HOST_IFMA_FORM");
    return e && atoi(e) == 4 ? 4 : 3;
  }();
  return form;
}

// 8 values, limb k of all of them in l[k]
struct alignas(64) V {
  uint64_t l[5][8];
};

inline void split52(const uint64_t v[4], uint64_t out[5]) {
  out[0] = v[0] & kMask52;
  out[1] = ((v[0] >> 52) | (v[1] << 12)) & kMask52;
  out[2] = ((v[1] >> 40) | (v[2] << 24)) & kMask52;
  out[3] = ((v[2] >> 28) | (v[3] << 36)) & kMask52;
  out[4] = v[3] >> 16;
}
inline void join52(const uint64_t in[5], uint64_t v[4]) {  // limbs below 2^52, value below 2^256
  v[0] = in[0] | (in[1] << 52);
  v[1] = (in[1] >> 12) | (in[2] << 40);
  v[2] = (in[2] >> 24) | (in[3] << 28);
  v[3] = (in[3] >> 36) | (in[4] << 16);
}

// Fr (a 2^256 mod r) -> the limbs of a 2^260 mod r: four modular doublings
inline void fr_to_limbs(const Fr& x, uint64_t out[5]) {
  Fr y = x + x;
  y = y + y;
  y = y + y;
  y = y + y;
  split52(y.v, out);
}
// limbs of (a 2^260 mod r) + multiples of r, value below 2^256 -> Fr
inline Fr fr_from_limbs(const uint64_t in[5]) {
  static const Fr inv16 = [] {
    Fr s = Fr::from_u64(16), i;
    s.invert(&i);
    return i;
  }();
  Fr t;
  join52(in, t.v);
  // below 4 r: the canonical residue by at most three subtractions
  for (int k = 0; k < 3; ++k) {
    bool lt = false;
    for (int i = 3; i >= 0; --i)
      if (t.v[i] != Fr::MOD[i]) {
        lt = t.v[i] < Fr::MOD[i];
        break;
      }
    if (lt) break;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 d = (unsigned __int128)t.v[i] - Fr::MOD[i] - (uint64_t)br;
      t.v[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }
  return t * inv16;  // t reads as the Montgomery form of 16 a
}

inline void set_lane(V& v, int lane, const Fr& x) {
  uint64_t l[5];
  fr_to_limbs(x, l);
  for (int k = 0; k < 5; ++k) v.l[k][lane] = l[k];
}
inline V zero() {
  V v;
  memset(&v, 0, sizeof v);
  return v;
}

// The tables of one parameter set in the lane layout (built once from the scalar schedule's tables).
struct Tables {
  int t = 0, r_f = 0, r_p = 0;
  V p;                      // the modulus in every lane
  V one;                    // 2^260 mod r in every lane (the Montgomery form of 1)
  uint64_t np = 0;          // -r^-1 mod 2^52
  V pre;                    // constants added with the input
  std::vector<V> full_k;    // post-S-box constants of the full rounds
  std::vector<V> mds_col, pre_sparse_col;  // column j of the matrix: lane i = M[i][j]
  std::vector<V> partial_k;  // lane 0 only
  std::vector<V> row;        // lane j = row[j]
  std::vector<V> col;        // lane 0 = 0, lane i >= 1 = col_hat[i - 1]
  // the three-product form of a partial round (t <= 7: lane 7 is free for the square), see `permute`
  std::vector<V> cvec;      // lane 0 = row[0], lane i >= 1 = col_hat[i - 1]
  std::vector<V> ck;        // cvec * partial_k  (the round's constant, already multiplied through)
  std::vector<V> row_rest;  // row with lane 0 cleared
};

struct Acc {  // ten 64-bit column accumulators per lane
  __m512i t[10];
};

SNARKV_IFMA void acc_zero(Acc& a) {
  for (int i = 0; i < 10; ++i) a.t[i] = _mm512_setzero_si512();
}
SNARKV_IFMA void load(const V& v, __m512i x[5]) {
  for (int k = 0; k < 5; ++k) x[k] = _mm512_load_si512((const void*)v.l[k]);
}
SNARKV_IFMA void store(V& v, const __m512i x[5]) {
  for (int k = 0; k < 5; ++k) _mm512_store_si512((void*)v.l[k], x[k]);
}
// a.t += x * y (lane-wise 260 x 260 -> 520 bits as column sums)
SNARKV_IFMA void acc_mul(Acc& a, const __m512i x[5], const __m512i y[5]) {
#pragma GCC unroll 5
  for (int i = 0; i < 5; ++i) {
#pragma GCC unroll 5
    for (int j = 0; j < 5; ++j) {
      a.t[i + j] = _mm512_madd52lo_epu64(a.t[i + j], x[i], y[j]);
      a.t[i + j + 1] = _mm512_madd52hi_epu64(a.t[i + j + 1], x[i], y[j]);
    }
  }
}
// a.t += x 2^260 (x rides through the reduction unchanged)
SNARKV_IFMA void acc_shifted(Acc& a, const __m512i x[5]) {
  for (int k = 0; k < 5; ++k) a.t[5 + k] = _mm512_add_epi64(a.t[5 + k], x[k]);
}
// Montgomery reduction of the column sums: out = a / 2^260 mod r (+ r at most), limbs carried below 2^52.
// Step i clears column i with m = t_i np mod 2^52 and hands its carry up.  The next step's m waits for column i + 1, i.e.
// for  t_{i+1} + lo(m p_1) + hi(m p_0) + carry(t_i + lo(m p_0)) : the two products go into SEPARATE registers and are
// joined by additions (4 + 2 cycles after m instead of two chained fused multiply-adds and an addition, 8 + 2); the
// remaining eight products of the step accumulate in place, off that path.  A sponge is one dependency chain of these.
SNARKV_IFMA void reduce(Acc& a, const __m512i p[5], __m512i np, __m512i out[5]) {
  const __m512i zero = _mm512_setzero_si512(), mask = _mm512_set1_epi64((long long)kMask52);
#pragma GCC unroll 5
  for (int i = 0; i < 5; ++i) {
    const __m512i m = _mm512_madd52lo_epu64(zero, a.t[i], np);  // (low 52 bits of t_i) * np mod 2^52
    const __m512i ti = _mm512_madd52lo_epu64(a.t[i], m, p[0]);  // 0 mod 2^52: only its carry lives on
    const __m512i u = _mm512_madd52lo_epu64(a.t[i + 1], m, p[1]);
    const __m512i v = _mm512_madd52hi_epu64(zero, m, p[0]);
    a.t[i + 1] = _mm512_add_epi64(_mm512_add_epi64(u, v), _mm512_srli_epi64(ti, 52));
#pragma GCC unroll 4
    for (int j = 1; j < 5; ++j) {
      if (j >= 2) a.t[i + j] = _mm512_madd52lo_epu64(a.t[i + j], m, p[j]);
      a.t[i + j + 1] = _mm512_madd52hi_epu64(a.t[i + j + 1], m, p[j]);
    }
  }
  __m512i c = zero;
#pragma GCC unroll 5
  for (int k = 0; k < 5; ++k) {
    const __m512i s = _mm512_add_epi64(a.t[5 + k], c);
    out[k] = k < 4 ? _mm512_and_si512(s, mask) : s;
    c = _mm512_srli_epi64(s, 52);
  }
}
// a.t = x * y, the low and the high halves of the 52 x 52 products summed in separate registers and joined at the end:
// a column's dependent chain is 5 fused multiply-adds instead of 9-10 (a product ON a dependency chain -- the S-box --
// is latency, not throughput)
SNARKV_IFMA void acc_mul_fresh(Acc& a, const __m512i x[5], const __m512i y[5]) {
  const __m512i zero = _mm512_setzero_si512();
  __m512i lo[9], hi[9];
#pragma GCC unroll 9
  for (int c = 0; c < 9; ++c) {
    bool first = true;
#pragma GCC unroll 5
    for (int i = 0; i < 5; ++i) {
      const int j = c - i;
      if (j < 0 || j > 4) continue;
      lo[c] = _mm512_madd52lo_epu64(first ? zero : lo[c], x[i], y[j]);
      hi[c] = _mm512_madd52hi_epu64(first ? zero : hi[c], x[i], y[j]);
      first = false;
    }
  }
  a.t[0] = lo[0];
#pragma GCC unroll 8
  for (int c = 1; c < 9; ++c) a.t[c] = _mm512_add_epi64(lo[c], hi[c - 1]);
  a.t[9] = hi[8];
}
SNARKV_IFMA void mul(const __m512i x[5], const __m512i y[5], const __m512i p[5], __m512i np, __m512i out[5]) {
  Acc a;
  acc_mul_fresh(a, x, y);
  reduce(a, p, np, out);
}
// x^5 + k (k may be null): the constant rides in the last product's accumulators
SNARKV_IFMA void sbox(const __m512i x[5], const __m512i* k, const __m512i p[5], __m512i np, __m512i out[5]) {
  __m512i x2[5], x4[5];
  mul(x, x, p, np, x2);
  mul(x2, x2, p, np, x4);
  Acc a;
  acc_zero(a);
  acc_mul(a, x4, x);
  if (k) acc_shifted(a, k);
  reduce(a, p, np, out);
}
SNARKV_IFMA void bcast(const __m512i x[5], int lane, __m512i out[5]) {
  const __m512i idx = _mm512_set1_epi64(lane);
  for (int k = 0; k < 5; ++k) out[k] = _mm512_permutexvar_epi64(idx, x[k]);
}
// lane 0 <- the sum of lanes 0 .. t-1 (the other lanes are zero by construction of the operands), lanes >= 1 <- 0
SNARKV_IFMA __m512i hsum_to_lane0(__m512i x) {
  const __m512i s = _mm512_set1_epi64((long long)_mm512_reduce_add_epi64(x));
  return _mm512_maskz_mov_epi64(0x01, s);
}

// s <- M s for the matrix given by its columns: t lane-wise products, one reduction
SNARKV_IFMA void dense(__m512i s[5], const V* cols, int t, const __m512i p[5], __m512i np) {
  Acc a, b;  // columns alternate between two sets of accumulators: half the dependent chain per register
  acc_zero(a);
  acc_zero(b);
  for (int j = 0; j < t; ++j) {
    __m512i c[5], bc[5];
    load(cols[j], c);
    bcast(s, j, bc);
    acc_mul((j & 1) ? b : a, c, bc);
  }
  for (int i = 0; i < 10; ++i) a.t[i] = _mm512_add_epi64(a.t[i], b.t[i]);
  reduce(a, p, np, s);
}

// state <- permutation(state); `state` holds the t words in lanes 0 .. t-1, limbs carried, values below 4 r on entry AND on
// exit (inside the partial rounds up to ~9.5 r < 2^260: header)
inline SNARKV_IFMA_FN void permute(V& state_v, const Tables& T) {
  const int t = T.t, h = T.r_f / 2;
  __m512i p[5], s[5];
  load(T.p, p);
  load(state_v, s);
  const __m512i np = _mm512_set1_epi64((long long)T.np);
  const __mmask8 live = (__mmask8)((1u << t) - 1);
  {  // s += pre, carried (the sum of two carried values: one carry pass)
    __m512i k[5];
    load(T.pre, k);
    const __m512i mask = _mm512_set1_epi64((long long)kMask52);
    __m512i c = _mm512_setzero_si512();
    for (int i = 0; i < 5; ++i) {
      const __m512i v = _mm512_add_epi64(_mm512_add_epi64(s[i], k[i]), c);
      s[i] = i < 4 ? _mm512_and_si512(v, mask) : v;
      c = _mm512_srli_epi64(v, 52);
    }
  }
  size_t fk = 0;
  for (int r = 0; r < h; ++r) {
    __m512i k[5];
    load(T.full_k[fk++], k);
    sbox(s, k, p, np, s);
    dense(s, (r + 1 < h ? T.mds_col : T.pre_sparse_col).data(), t, p, np);
  }
  if (t <= 7 && partial_round_form() == 3) {
    // A partial round is  x <- s_0^5 + k ;  s_0 <- row . s ;  s_i <- s_i + col_i x  (i >= 1): on a chain, the three
    // products of the S-box and then the row's.  With c = (row_0, col_1 .. col_{t-1}) the new state is
    //     c x^5  +  [ c k  +  (sum_{j >= 1} row_j s_j ,  s_1 .. s_{t-1}) ]        and   c x^5 = (c x) x^4 :
    //   P1 = bcast(s_0) * (c | s_0 in lane 7)       lanes 0 .. t-1: c_i s_0, lane 7: s_0^2                          product 1
    //   M2 = (s_0^2 | s_1 ..) * (s_0^2 | row_1 ..)  lane 0: s_0^4, lanes j >= 1: row_j s_j -- the row's products ride in
    //                                               the lanes the fourth power leaves idle, reduced with it            product 2
    //   s  = P1 * bcast(M2[0]) + the bracket        (the sum of M2's lanes 1 .. t-1 enters as a 2^260-shifted addend)    product 3
    // THREE dependent products per round instead of four, and 315 fused multiply-adds instead of 470.
    // Bounds: lane 0 comes out below 8 r (a product, c_0 k, four reduced row products, the reduction's + r); words
    // 1 .. t-1 gain up to 2 r per round (c_i k rides along with s_i) and are brought back below 1.3 r every SIXTEENTH
    // round by a product with 1 that the chain does not wait for (lane 0 skips it): 1.3 r + 16 x 2 r < 34 r < 2^260 / r = 64,
    // so every limb stays below 2^52, and as an operand such a word only meets a table entry below r (34 r^2 / 2^260 < r).
    // The full rounds that follow multiply every word (x^2 of 34 r is below 14 r), so the permutation still ends below
    // 4 r.  Lane 7 carries a by-product (s_0^6) that nothing reads; the final store clears it.
    for (int r = 0; r < T.r_p; ++r) {
      __m512i bx[5], y[5], c[5], p1[5], b2[5], a2[5], y2[5], m2[5], x4[5], rr[5], ck[5];
      bcast(s, 0, bx);
      load(T.cvec[(size_t)r], c);
      for (int i = 0; i < 5; ++i) y[i] = _mm512_mask_mov_epi64(c[i], 0x80, bx[i]);
      mul(bx, y, p, np, p1);
      bcast(p1, 7, b2);
      load(T.row_rest[(size_t)r], rr);
      for (int i = 0; i < 5; ++i) {
        a2[i] = _mm512_mask_mov_epi64(s[i], 0x01, b2[i]);
        y2[i] = _mm512_mask_mov_epi64(rr[i], 0x01, b2[i]);
      }
      mul(a2, y2, p, np, m2);
      bcast(m2, 0, x4);
      load(T.ck[(size_t)r], ck);
      Acc a;
      acc_mul_fresh(a, p1, x4);
      for (int i = 0; i < 5; ++i) {
        const __m512i rows = hsum_to_lane0(_mm512_maskz_mov_epi64(0xFE, m2[i]));  // sum_{j >= 1} row_j s_j (each reduced), into lane 0
        const __m512i ride = _mm512_add_epi64(ck[i], _mm512_maskz_mov_epi64((__mmask8)(live & ~1u), s[i]));
        a.t[5 + i] = _mm512_add_epi64(a.t[5 + i], _mm512_add_epi64(ride, rows));
      }
      reduce(a, p, np, s);
      if ((r & 15) == 15) {
        __m512i one[5], n[5];
        load(T.one, one);
        mul(s, one, p, np, n);
        for (int i = 0; i < 5; ++i) s[i] = _mm512_mask_mov_epi64(n[i], 0x01, s[i]);
      }
    }
  } else
  for (int r = 0; r < T.r_p; ++r) {
    __m512i k[5], sb[5], row[5], col[5], b0[5];
    load(T.partial_k[(size_t)r], k);
    sbox(s, k, p, np, sb);  // (every lane computes; only lane 0 is kept)
    for (int i = 0; i < 5; ++i) s[i] = _mm512_mask_mov_epi64(s[i], 0x01, sb[i]);
    load(T.row[(size_t)r], row);
    load(T.col[(size_t)r], col);
    Acc d;  // the dot product row . state, lane by lane, then across the lanes into lane 0
    acc_zero(d);
    acc_mul(d, row, s);
    bcast(s, 0, b0);
    Acc a;  // lanes i >= 1: col_hat[i - 1] s0 + state_i 2^260; lane 0: the dot product
    acc_zero(a);
    acc_mul(a, col, b0);
    for (int i = 0; i < 5; ++i) a.t[5 + i] = _mm512_add_epi64(a.t[5 + i], _mm512_maskz_mov_epi64((__mmask8)(live & ~1u), s[i]));
    for (int i = 0; i < 10; ++i) a.t[i] = _mm512_add_epi64(a.t[i], hsum_to_lane0(d.t[i]));
    reduce(a, p, np, s);
    // words 1 .. t-1 pass through the reduction as value 2^260 and pick up to + r per round (nothing subtracts here):
    // every eighth round a product with 1 (= 2^260 mod r) brings every word back below 1.3 r
    if ((r & 7) == 7) {
      __m512i one[5];
      load(T.one, one);
      mul(s, one, p, np, s);
    }
  }
  for (int r = 0; r < h; ++r) {
    const bool last = r + 1 == h;
    if (last) {
      sbox(s, nullptr, p, np, s);
    } else {
      __m512i k[5];
      load(T.full_k[fk++], k);
      sbox(s, k, p, np, s);
    }
    dense(s, T.mds_col.data(), t, p, np);
  }
  for (int i = 0; i < 5; ++i) s[i] = _mm512_maskz_mov_epi64(live, s[i]);
  store(state_v, s);
}

// state lanes 1 .. n += inputs, lane n + 1 += 1 (the sponge's padding word, if it fits), carried
inline SNARKV_IFMA_FN void absorb(V& state_v, const V& addend) {
  __m512i s[5], k[5];
  load(state_v, s);
  load(addend, k);
  const __m512i mask = _mm512_set1_epi64((long long)kMask52);
  __m512i c = _mm512_setzero_si512();
  for (int i = 0; i < 5; ++i) {
    const __m512i v = _mm512_add_epi64(_mm512_add_epi64(s[i], k[i]), c);
    s[i] = i < 4 ? _mm512_and_si512(v, mask) : v;
    c = _mm512_srli_epi64(v, 52);
  }
  store(state_v, s);
}


// dev aid (tools / test hook): n DEPENDENT lane-wise products x <- x * x, the latency a sponge is made of
inline SNARKV_IFMA_FN void mul_chain(V& xv, const V& pv, uint64_t np64, size_t n) {
  __m512i p[5], x[5];
  load(pv, p);
  load(xv, x);
  const __m512i np = _mm512_set1_epi64((long long)np64);
  for (size_t i = 0; i < n; ++i) mul(x, x, p, np, x);
  store(xv, x);
}

// Lane-wise  y = (x^3 + b)^e  for canonical x (limbs of plain integers below the modulus), the result as PLAIN integers
// below 4 p (limbs carried): the vector half of a batch of point decompressions (transcript.hpp `g1_decompress_x8`).
// `e` = four 64-bit words (below 2^253 for BN254's (p + 1) / 4); r2 = R^2, one_m = R, b_m = b R (mod p, R = 2^260).
inline SNARKV_IFMA_FN void sqrt_x3_plus_b(const V& x_plain, const V& pv, uint64_t np64, const V& r2v, const V& one_mv, const V& b_mv,
                                          const V& one_plainv, const uint64_t e[4], V& out) {
  __m512i p[5], x[5], r2[5], xm[5], x2[5], a[5], bm[5];
  load(pv, p);
  load(x_plain, x);
  load(r2v, r2);
  load(b_mv, bm);
  const __m512i np = _mm512_set1_epi64((long long)np64), mask = _mm512_set1_epi64((long long)kMask52);
  mul(x, r2, p, np, xm);  // x R
  mul(xm, xm, p, np, x2);
  {
    Acc t;  // x^3 + b: the constant rides in the product's accumulators
    acc_mul_fresh(t, x2, xm);
    acc_shifted(t, bm);
    reduce(t, p, np, a);
  }
  // 4-bit fixed window: table[k] = a^k, k = 0 .. 15 (in memory: 16 x 5 registers do not fit the file)
  V table[16];
  {
    __m512i one[5], cur[5];
    load(one_mv, one);
    store(table[0], one);
    store(table[1], a);
    for (int k = 0; k < 5; ++k) cur[k] = a[k];
    for (int k = 2; k < 16; ++k) {
      mul(cur, a, p, np, cur);
      store(table[k], cur);
    }
  }
  __m512i acc[5];
  bool started = false;
  for (int w = 63; w >= 0; --w) {
    const unsigned d = (unsigned)((e[w >> 4] >> (4 * (w & 15))) & 15u);
    if (started) {
      for (int q = 0; q < 4; ++q) mul(acc, acc, p, np, acc);
      if (d) {
        __m512i tk[5];
        load(table[d], tk);
        mul(acc, tk, p, np, acc);
      }
    } else if (d) {
      load(table[d], acc);
      started = true;
    }
  }
  if (!started) load(one_mv, acc);
  __m512i onep[5], y[5];
  load(one_plainv, onep);
  mul(acc, onep, p, np, y);  // out of the Montgomery domain (below 1.3 p)
  (void)mask;
  store(out, y);
}

}  // namespace poseidon_ifma
}  // namespace snarkv_host
#else
namespace snarkv_host {
namespace poseidon_ifma {
struct Tables {};
inline bool available() { return false; }
}  // namespace poseidon_ifma
}  // namespace snarkv_host
#endif
