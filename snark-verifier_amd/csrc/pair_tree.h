// Batched-affine PAIR LEVEL in front of the bucket accumulation (msm_pippenger.hip, P3b).
//
// k_accumulate adds every sorted entry into its bucket with the XYZZ mixed addition: 10 field products (8M + 2S, 9
// reductions) per entry -- the floor of that formulation.  Two AFFINE points can be added with
// 3 products (lambda = (y1 - y0) / (x1 - x0), x3 = lambda^2 - x0 - x1, y3 = lambda (x0 - x3) - y0) once 1 / (x1 - x0) is
// known, and a whole level of independent inversions costs 3 products each by Montgomery's trick (one running product
// forward, two on the way back) plus ONE real inversion for all of them: 6 products per addition instead of 10.
// The bucket-sorted stream makes the level's pairs free: with every bucket's entry count padded to even
// (k_sort_level2, pad mode) the entries (2i, 2i + 1) always share a bucket, so pair slot i = their sum is one entry of
// a HALF-length stream that the unchanged XYZZ accumulation then consumes.  Per MSM at 2^20 points: 8.4 M affine
// additions at 6 products replace 8.4 M of the 16.8 M mixed additions at 10.
//
//   forward  (k_pair_fwd)   lane j walks its m slots:  P_k = d_0 ... d_(k-1) stored, total_j = d_0 ... d_(m-1)
//   invert   (k_binv_*)     the lane totals are inverted by the same trick, recursively (m2 per lane), down to <= 1024
//                           values that one workgroup inverts with an LDS product tree and ONE safegcd inversion
//   backward (k_pair_bwd)   lane j walks its slots in reverse:  1/d_k = I P_k,  I <- I d_k ; the affine sum is stored as
//                           2 x 9 lazy limbs (72 B), which the accumulation reads without the 256-bit unpack
//
// Slot kinds.  d must never be 0, so the exceptional pairs are classified on the CANONICAL level-1 inputs (word
// equality): pad / copy -> d = 1;  P + P -> d = 2y, lambda = 3x^2 / 2y;  P + (-P) -> d = 1, result = identity (a SKIP
// entry of the half-length stream).
//
// Everything here is per-lane code with explicit indices (no threadIdx): the kernels call it with their lane id, and
// tests/hosttest runs the same functions in host loops against the big-integer oracle.
#pragma once
#include <stddef.h>
#include "g1_29.h"

namespace snarkv {

constexpr uint32_t kEntrySkip = 0x40000000u;  // bit 30 of an entry's .y: pad / cancelled pair -- contributes nothing
constexpr uint32_t kEntryIdx = 0x3FFFFFFFu;   // point index bits of an entry's .y (bit 31 = negate)

struct PairEntry {  // layout of the sorted stream's uint2: .x = bucket id, .y = point index | skip << 30 | neg << 31
  uint32_t bucket, y;
};

enum PairKind : uint32_t { PAIR_SKIP = 0, PAIR_COPY = 1, PAIR_ADD = 2, PAIR_DBL = 3, PAIR_CANCEL = 4 };

// lane j of `lanes`, k-th of its m elements, workgroups of T lanes: consecutive lanes take consecutive elements, so
// every load / store of a step is one contiguous segment per wavefront
SNARKV_HD size_t pair_elem(uint32_t j, uint32_t T, uint32_t m, uint32_t k) {
  return ((size_t)(j / T) * m + k) * T + (j % T);
}

// strided (structure-of-arrays) field element storage: limb l of element e at a[l * stride + e]
SNARKV_HD Fq29 soa_load(const int32_t* a, size_t stride, size_t e) {
  Fq29 r;
#pragma unroll
  for (int l = 0; l < 9; ++l) r.v[l] = a[(size_t)l * stride + e];
  return r;
}
SNARKV_HD void soa_store(int32_t* a, size_t stride, size_t e, const Fq29& v) {
#pragma unroll
  for (int l = 0; l < 9; ++l) a[(size_t)l * stride + e] = v.v[l];
}

SNARKV_HD bool words8_equal(const uint32_t* a, const uint32_t* b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d |= a[i] ^ b[i];
  return d == 0;
}

// classification of pair slot (e0, e1) and its denominator d (never 0).  x0 / x1: the 8 canonical words of the two x
// coordinates (loaded by the caller when both entries are real); y0 / y1 likewise, read only when x0 == x1.
SNARKV_HD uint32_t pair_kind_of_flags(const PairEntry& e0, const PairEntry& e1) {
  if (e0.y & kEntrySkip) return PAIR_SKIP;   // pads follow the real entries of a bucket: (pad, real) never occurs
  if (e1.y & kEntrySkip) return PAIR_COPY;
  return PAIR_ADD;                           // refined by pair_refine once the coordinates are known
}
SNARKV_HD uint32_t pair_refine(const PairEntry& e0, const PairEntry& e1, const uint32_t* x0, const uint32_t* x1,
                               const uint32_t* y0, const uint32_t* y1) {
  if (!words8_equal(x0, x1)) return PAIR_ADD;
  // same x: the points are equal or opposite (y != 0 on a prime-order curve); the raw y words decide together with the signs
  const bool same_raw = words8_equal(y0, y1);
  const bool same_sign = ((e0.y ^ e1.y) >> 31) == 0;
  return same_raw == same_sign ? PAIR_DBL : PAIR_CANCEL;
}

// the denominator of an ADD / DBL slot from unpacked limbs (y0 already carries its sign)
SNARKV_HD Fq29 pair_denominator(uint32_t kind, const Fq29& x0, const Fq29& x1, const Fq29& y0s) {
  if (kind == PAIR_ADD) return fq29_sub(x1, x0);  // |limb| < 2^29
  if (kind == PAIR_DBL) return fq29_norm(fq29_dbl(y0s));  // carry-normalised: it may meet a raw difference in a product
  return fq29_one();
}

// the affine sum of an ADD / DBL slot given inv = 1 / d:  lazy limbs, carry-normalised
SNARKV_HD void pair_sum(uint32_t kind, const Fq29& x0, const Fq29& y0s, const Fq29& x1, const Fq29& y1s, const Fq29& inv,
                        Fq29& x3, Fq29& y3) {
  Fq29 num;
  if (kind == PAIR_DBL) {
    Fq29 xx = fq29_sqr(x0);                       // canonical input: carry-normalised
    num = fq29_norm(fq29_add(fq29_dbl(xx), xx));  // 3 x^2
  } else {
    num = fq29_sub(y1s, y0s);                     // |limb| < 2^30
  }
  Fq29 lam = fq29_mul(inv, num);                  // inv: product output (limbs < 2^29)
  Fq29 ll = fq29_sqr(lam);
  x3 = fq29_norm(fq29_sub(fq29_sub(ll, x0), x1));  // limbs before norm in (-2^30, 2^29); value in (-2.2 p, 1.2 p)
  Fq29 t = fq29_sub(x0, x3);                       // |limb| < 2^29
  y3 = fq29_norm(fq29_sub(fq29_mul(lam, t), y0s)); // value in (-1.2 p, 2.2 p)
}

// ------------------------------------------------------------------------------------------------ level kernels' lanes
// Geometry of one level: `nslots` live pair slots (device value), S = slot capacity (strides of the SoA arrays),
// lanes = ceil(S / m) rounded up to whole workgroups of T.

// forward: prefix products of the lane's denominators.  pfx[k > 0] = d_0 .. d_(k-1) at the slot's index; tot[j] = all m.
SNARKV_HD void pair_fwd_lane(uint32_t j, uint32_t T, uint32_t m, uint32_t nslots, const PairEntry* entries,
                             const G1Packed* pts, int32_t* pfx, size_t S, int32_t* tot, size_t L) {
  Fq29 pr = fq29_one();
  bool have = false;  // pr holds at least one real denominator
  for (uint32_t k = 0; k < m; ++k) {
    const size_t i = pair_elem(j, T, m, k);
    if (i >= nslots) break;  // slots are handed out in increasing order over k: nothing live beyond
    if (k > 0) soa_store(pfx, S, i, pr);
    const PairEntry e0 = entries[2 * i], e1 = entries[2 * i + 1];
    uint32_t kind = pair_kind_of_flags(e0, e1);
    if (kind != PAIR_ADD) continue;  // pad / copy: d = 1, the running product is unchanged
    const G1Packed& a = pts[e0.y & kEntryIdx];
    const G1Packed& b = pts[e1.y & kEntryIdx];
    kind = pair_refine(e0, e1, a.w, b.w, a.w + 8, b.w + 8);
    if (kind == PAIR_CANCEL) continue;
    Fq29 x0 = fq29_unpack256(a.w), x1 = fq29_unpack256(b.w), y0 = fq29_zero();
    if (kind == PAIR_DBL) {
      y0 = fq29_unpack256(a.w + 8);
      if (e0.y >> 31) y0 = fq29_neg(y0);
    }
    const Fq29 d = pair_denominator(kind, x0, x1, y0);
    pr = have ? fq29_mul(pr, d) : d;  // a raw difference (|limb| < 2^29) is a valid product operand
    have = true;
  }
  soa_store(tot, L, j, fq29_norm(pr));
}

// backward: itot[j] = 1 / tot[j].  Writes the half-length stream: out_entries[i] = {bucket, i | skip}, out_pts[18 i ..] =
// x3 | y3 as lazy limbs.
SNARKV_HD void pair_bwd_lane(uint32_t j, uint32_t T, uint32_t m, uint32_t nslots, const PairEntry* entries,
                             const G1Packed* pts, const int32_t* pfx, size_t S, const int32_t* itot, size_t L,
                             PairEntry* out_entries, int32_t* out_pts) {
  // the lane's live slots are k = 0 .. last
  int last = -1;
  for (uint32_t k = 0; k < m; ++k)
    if (pair_elem(j, T, m, k) < nslots) last = (int)k;
  if (last < 0) return;
  Fq29 I = soa_load(itot, L, j);
  for (int k = last; k >= 0; --k) {
    const size_t i = pair_elem(j, T, m, (uint32_t)k);
    const PairEntry e0 = entries[2 * i], e1 = entries[2 * i + 1];
    uint32_t kind = pair_kind_of_flags(e0, e1);
    PairEntry oe;
    oe.bucket = e0.bucket;
    oe.y = (uint32_t)i;
    Fq29 x3 = fq29_zero(), y3 = fq29_zero(), d = fq29_one();
    if (kind == PAIR_SKIP) {
      oe.y |= kEntrySkip;
    } else {
      const G1Packed& a = pts[e0.y & kEntryIdx];
      Fq29 x0 = fq29_unpack256(a.w), y0 = fq29_unpack256(a.w + 8);
      if (e0.y >> 31) y0 = fq29_neg(y0);
      if (kind == PAIR_COPY) {
        x3 = x0;
        y3 = y0;
      } else {
        const G1Packed& b = pts[e1.y & kEntryIdx];
        kind = pair_refine(e0, e1, a.w, b.w, a.w + 8, b.w + 8);
        Fq29 x1 = fq29_unpack256(b.w), y1 = fq29_unpack256(b.w + 8);
        if (e1.y >> 31) y1 = fq29_neg(y1);
        if (kind == PAIR_CANCEL) {
          oe.y |= kEntrySkip;
        } else {
          d = pair_denominator(kind, x0, x1, y0);
          Fq29 inv = I;
          if (k > 0) inv = fq29_mul(I, soa_load(pfx, S, i));
          pair_sum(kind, x0, y0, x1, y1, inv, x3, y3);
        }
      }
    }
    if (k > 0 && (kind == PAIR_ADD || kind == PAIR_DBL)) I = fq29_mul(I, d);  // d = 1 otherwise
    out_entries[i] = oe;
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      out_pts[18 * i + l] = x3.v[l];
      out_pts[18 * i + 9 + l] = y3.v[l];
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused form (pair RUNS)
// The level above writes its sums out (72 B per slot) and a second kernel accumulates them.  The fused form keeps the
// run structure of the bucket accumulation instead: lane t owns RUN consecutive entries of the PADDED stream (RUN even:
// H = RUN / 2 pair slots), the forward kernel leaves the prefix products of its slots' denominators, and the backward
// kernel walks the same slots in reverse, forms each affine pair sum and adds it straight into the lane's XYZZ bucket
// accumulator -- head / tail partials and interior buckets exactly as k_accumulate leaves them for k_combine.  No sums
// are written or re-read, and the half-length stream never exists: 3.0 GB through the fabric per MSM instead of 4.3.
// Prefix scratch is private to the two kernels: element (wavefront, slot k, limb, lane) -- every store one segment.
constexpr uint32_t kPairNoBucket = 0xFFFFFFFFu;  // == kNoBucket of msm_pippenger.hip

SNARKV_HD size_t pairrun_pfx_index(uint32_t t, uint32_t H, uint32_t k, int limb) {
  return ((((size_t)(t >> 6) * H + k) * 9 + (size_t)limb) << 6) + (t & 63u);
}

// A slot as the kernels hold it between its loads and its arithmetic: the two entries and the raw (packed) words of the
// two points.  Loading is separate from using so that slot k - 1 (backward) / k + 1 (forward) is in flight while slot k's
// ~3 500 instructions run -- the same software pipeline as k_accumulate's.
struct PairRaw {
  PairEntry e0, e1;
  uint32_t a[16], b[16];  // x | y words of e0's / e1's point (stale when the entry is a pad)
};
template <bool NEED_Y>
SNARKV_HD void pair_raw_load(PairRaw& r, const PairEntry* entries, size_t i0, const G1Packed* pts) {
  r.e0 = entries[i0];
  r.e1 = entries[i0 + 1];
  const uint32_t ia = (r.e0.y & kEntrySkip) ? 0u : (r.e0.y & kEntryIdx);  // a pad reads point 0: always mapped, never used
  const uint32_t ib = (r.e1.y & kEntrySkip) ? ia : (r.e1.y & kEntryIdx);
  const uint32_t* pa = pts[ia].w;
  const uint32_t* pb = pts[ib].w;
#pragma unroll
  for (int j = 0; j < (NEED_Y ? 16 : 8); ++j) {
    r.a[j] = pa[j];
    r.b[j] = pb[j];
  }
}

// what a slot contributes: kind + (for ADD / DBL) the operands; y carry their signs
struct PairSlot {
  uint32_t kind;
  Fq29 x0, y0, x1, y1;
};
// NEED_Y = false (the forward pass): only what the denominator needs; the y words are then read from memory in the rare
// x0 == x1 case (`pts`), not from the raw slot
template <bool NEED_Y>
SNARKV_HD PairSlot pair_slot_decode(const PairRaw& r, const G1Packed* pts) {
  PairSlot s;
  s.kind = pair_kind_of_flags(r.e0, r.e1);
  s.x0 = s.y0 = s.x1 = s.y1 = fq29_zero();
  if (s.kind == PAIR_SKIP) return s;
  if (s.kind == PAIR_COPY) {
    if (NEED_Y) {
      s.x0 = fq29_unpack256(r.a);
      s.y0 = fq29_unpack256(r.a + 8);
      if (r.e0.y >> 31) s.y0 = fq29_neg(s.y0);
    }
    return s;
  }
  const uint32_t* ya = NEED_Y ? r.a + 8 : pts[r.e0.y & kEntryIdx].w + 8;
  const uint32_t* yb = NEED_Y ? r.b + 8 : pts[r.e1.y & kEntryIdx].w + 8;
  s.kind = pair_refine(r.e0, r.e1, r.a, r.b, ya, yb);
  if (s.kind == PAIR_CANCEL) return s;
  s.x0 = fq29_unpack256(r.a);
  s.x1 = fq29_unpack256(r.b);
  if (NEED_Y || s.kind == PAIR_DBL) {
    s.y0 = fq29_unpack256(ya);
    if (r.e0.y >> 31) s.y0 = fq29_neg(s.y0);
  }
  if (NEED_Y) {
    s.y1 = fq29_unpack256(yb);
    if (r.e1.y >> 31) s.y1 = fq29_neg(s.y1);
  }
  return s;
}

// forward: lane t, entries [t RUN, min((t + 1) RUN, stop)); `stop` = padded entries (even).  tot[t] = the product of the
// lane's denominators (1 for a lane beyond the stream: the inversion levels run over every lane)
SNARKV_HD void pairrun_fwd_lane(uint32_t t, uint32_t RUN, uint32_t stop, const PairEntry* entries, const G1Packed* pts,
                                int32_t* pfx, int32_t* tot, size_t L) {
  const uint64_t begin = (uint64_t)t * RUN;
  const uint32_t H = RUN / 2;
  Fq29 pr = fq29_one();
  bool have = false;
  if (begin < stop) {
    const uint32_t nsl = (uint32_t)(((stop - begin > RUN) ? RUN : (stop - begin)) / 2);
    PairRaw cur, nxt;
    pair_raw_load<false>(nxt, entries, begin, pts);
    for (uint32_t k = 0; k < nsl; ++k) {
      cur = nxt;
      if (k + 1 < nsl) pair_raw_load<false>(nxt, entries, begin + 2 * (k + 1), pts);  // in flight during slot k's product
      if (k > 0) {
#pragma unroll
        for (int l = 0; l < 9; ++l) pfx[pairrun_pfx_index(t, H, k, l)] = pr.v[l];
      }
      const PairSlot sl = pair_slot_decode<false>(cur, pts);
      if (sl.kind != PAIR_ADD && sl.kind != PAIR_DBL) continue;
      const Fq29 d = pair_denominator(sl.kind, sl.x0, sl.x1, sl.y0);
      pr = have ? fq29_mul(pr, d) : d;
      have = true;
    }
  }
  soa_store(tot, L, t, fq29_norm(pr));
}

// backward + accumulate: itot[t] = 1 / tot[t].  Leaves, for run slot t (as k_accumulate does): seg_ids[2t] / seg_parts[2t] =
// the partial of the run's FIRST bucket, [2t + 1] = of its LAST bucket if different (kPairNoBucket otherwise, and for a
// part that held skip slots only), complete interior buckets straight in `buckets`.
SNARKV_HD void pairrun_bwd_lane(uint32_t t, uint32_t RUN, uint32_t stop, const PairEntry* entries, const G1Packed* pts,
                                const int32_t* pfx, const int32_t* itot, size_t L, G1Xyzz29* buckets, uint32_t* seg_ids,
                                G1Xyzz29* seg_parts) {
  const uint64_t begin = (uint64_t)t * RUN;
  if (begin >= stop) return;
  const uint32_t H = RUN / 2;
  const uint32_t nsl = (uint32_t)(((stop - begin > RUN) ? RUN : (stop - begin)) / 2);
  Fq29 I = soa_load(itot, L, t);
  PairRaw cur, nxt;
  Fq29 pr_cur = fq29_one(), pr_nxt = fq29_one();
  auto load = [&](int k) {  // slot k's entries, points and prefix
    pair_raw_load<true>(nxt, entries, begin + 2 * (size_t)k, pts);
    if (k > 0) {
#pragma unroll
      for (int l = 0; l < 9; ++l) pr_nxt.v[l] = pfx[pairrun_pfx_index(t, H, (uint32_t)k, l)];
    }
  };
  load((int)nsl - 1);
  uint32_t cur_b = nxt.e0.bucket;
  bool tail_written = false, fresh = true;
  G1Xyzz29 acc = xyzz29_identity();
  for (int k = (int)nsl - 1; k >= 0; --k) {
    cur = nxt;
    pr_cur = pr_nxt;
    if (k > 0) load(k - 1);  // in flight during slot k's ~3 500 instructions
    if (cur.e0.bucket != cur_b) {  // walking down: the bucket above is finished
      if (!tail_written) {
        seg_ids[2 * (size_t)t + 1] = fresh ? kPairNoBucket : cur_b;
        if (!fresh) seg_parts[2 * (size_t)t + 1] = acc;
        tail_written = true;
      } else if (!fresh) {
        buckets[cur_b] = acc;  // complete interior bucket
      }
      cur_b = cur.e0.bucket;
      fresh = true;
    }
    const PairSlot sl = pair_slot_decode<true>(cur, pts);
    if (sl.kind == PAIR_SKIP || sl.kind == PAIR_CANCEL) continue;
    G1Affine29 p;
    if (sl.kind == PAIR_COPY) {
      p.x = sl.x0;
      p.y = sl.y0;
    } else {
      const Fq29 d = pair_denominator(sl.kind, sl.x0, sl.x1, sl.y0);
      Fq29 inv = I;
      if (k > 0) {
        inv = fq29_mul(I, pr_cur);
        I = fq29_mul(I, d);
      }
      pair_sum(sl.kind, sl.x0, sl.y0, sl.x1, sl.y1, inv, p.x, p.y);
    }
    if (fresh) {
      acc.x = p.x;
      acc.y = p.y;
      acc.zz = fq29_one();
      acc.zzz = fq29_one();
      fresh = false;
    } else {
      xyzz29_madd_fast(acc, p);
    }
  }
  seg_ids[2 * (size_t)t] = fresh ? kPairNoBucket : cur_b;
  if (!fresh) seg_parts[2 * (size_t)t] = acc;
  if (!tail_written) seg_ids[2 * (size_t)t + 1] = kPairNoBucket;
}

// ---- the same trick on plain arrays of field elements (the lane totals): a[0 .. N) -> inverses in place ----------------
// up: pfx[e] (k > 0) = product of the lane's elements before e; tot[j] = product of all
SNARKV_HD void binv_up_lane(uint32_t j, uint32_t T, uint32_t m, uint32_t N, const int32_t* a, size_t A, int32_t* pfx,
                            int32_t* tot, size_t L) {
  Fq29 pr = fq29_one();
  bool started = false;
  for (uint32_t k = 0; k < m; ++k) {
    const size_t e = pair_elem(j, T, m, k);
    if (e >= N) break;
    Fq29 d = soa_load(a, A, e);
    if (started) {
      soa_store(pfx, A, e, pr);
      pr = fq29_mul(pr, d);
    } else {
      pr = d;
      started = true;
    }
  }
  soa_store(tot, L, j, pr);  // products / stored totals are carry-normalised already
}
// down: itot[j] = 1 / tot[j]; a[e] <- 1 / a[e]
SNARKV_HD void binv_down_lane(uint32_t j, uint32_t T, uint32_t m, uint32_t N, int32_t* a, size_t A, const int32_t* pfx,
                              const int32_t* itot, size_t L) {
  int last = -1;
  for (uint32_t k = 0; k < m; ++k)
    if (pair_elem(j, T, m, k) < N) last = (int)k;
  if (last < 0) return;
  Fq29 I = soa_load(itot, L, j);
  for (int k = last; k >= 0; --k) {
    const size_t e = pair_elem(j, T, m, (uint32_t)k);
    if (k == 0) {
      soa_store(a, A, e, I);
    } else {
      Fq29 d = soa_load(a, A, e);
      soa_store(a, A, e, fq29_mul(I, soa_load(pfx, A, e)));
      I = fq29_mul(I, d);
    }
  }
}

}  // namespace snarkv
