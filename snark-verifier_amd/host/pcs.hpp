// Mirror of the reference's PCS / accumulation-scheme layer for KZG on the
// native loader, EC work on the MI355X:
//
//   Query, PolynomialCommitmentScheme,           snark-verifier/src/pcs.rs:21-184
//   AccumulationScheme, AccumulationDecider,
//   AccumulationSchemeProver, AccumulatorEncoding
//   KzgSuccinctVerifyingKey                      snark-verifier/src/pcs/kzg.rs:19-35
//   KzgAccumulator, LimbsEncoding                snark-verifier/src/pcs/kzg/accumulator.rs:6-81
//   KzgDecidingKey, decide / decide_all          snark-verifier/src/pcs/kzg/decider.rs:6-93
//   KzgAs (verify / read_proof / create_proof)   snark-verifier/src/pcs/kzg/accumulation.rs:17-197
//   Gwc19                                        snark-verifier/src/pcs/kzg/multiopen/gwc19.rs:21-160
//   Bdfg21                                       snark-verifier/src/pcs/kzg/multiopen/bdfg21.rs:27-371
//
// Rust traits become C++ templates over the multi-open scheme tag (Gwc19 /
// Bdfg21), exactly as `KzgAs<M, MOS>` is generic over `MOS`.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <variant>
#include <functional>

#include "msm.hpp"

#ifndef SNARKV_HOST_TRACE
#define SNARKV_HOST_TRACE 0  // build with -DSNARKV_HOST_TRACE=1 for the host / device split of the phases on stderr (dev aid)
#endif
namespace snarkv_host {

using L = GpuNativeLoader;
using MsmT = Msm<L>;

template <class T>
struct Result {
  std::optional<T> value;
  Error err;
  bool ok() const { return value.has_value(); }
  static Result Ok(T v) {
    Result r;
    r.value = std::move(v);
    return r;
  }
  static Result Err(Error e) {
    Result r;
    r.err = std::move(e);
    return r;
  }
};

// pcs.rs:21-48
template <class T = std::monostate>
struct Query {
  size_t poly;
  Fr shift;
  T eval;
};

// util/transcript.rs:9-62 -- only the interface; the hash-based transcripts
// (Poseidon, Keccak) are outside the hot path.
struct Transcript {
  virtual ~Transcript() = default;
  virtual Fr squeeze_challenge() = 0;
  virtual Error common_ec_point(const G1Affine& p) = 0;
  virtual Error common_scalar(const Fr& s) = 0;
  virtual Result<G1Affine> read_ec_point() = 0;
  virtual Result<Fr> read_scalar() = 0;
  virtual Error write_ec_point(const G1Affine& p) = 0;
  virtual Error write_scalar(const Fr& s) = 0;
  std::vector<Fr> squeeze_n_challenges(size_t n) {
    std::vector<Fr> v;
    for (size_t i = 0; i < n; ++i) v.push_back(squeeze_challenge());
    return v;
  }
  // called before `n` consecutive point reads: a transcript may decode them together (transcript.hpp: the Poseidon
  // transcript's compressed points on AVX-512 IFMA) -- an optimisation only, every read still checks what it takes
  virtual void prefetch_points(size_t) {}
  Result<std::vector<G1Affine>> read_n_ec_points(size_t n) {
    std::vector<G1Affine> v;
    if (n >= 2) prefetch_points(n);
    for (size_t i = 0; i < n; ++i) {
      auto r = read_ec_point();
      if (!r.ok()) return Result<std::vector<G1Affine>>::Err(r.err);
      v.push_back(*r.value);
    }
    return Result<std::vector<G1Affine>>::Ok(v);
  }
};

#if !defined(SNARKV_HOST_PALLAS)  // ---- KZG: BN254 only (pairing decider) ----
// kzg.rs:19-35
struct KzgSuccinctVerifyingKey {
  G1Affine g;
};

// accumulator.rs:6-26
struct KzgAccumulator {
  G1Affine lhs, rhs;
  void to_bytes(uint8_t out[128]) const {
    memcpy(out, lhs.b, 64);
    memcpy(out + 64, rhs.b, 64);
  }
};

// decider.rs:6-42.  The device-side G2 line tables are built once and cached
// (the reference redoes `G2Prepared::from` on every decide, decider.rs:74).
struct KzgDecidingKey {
  KzgSuccinctVerifyingKey svk;
  G2Affine g2, s_g2;
  KzgDecidingKey(const G1Affine& g1, const G2Affine& g2_, const G2Affine& s_g2_) : svk{g1}, g2(g2_), s_g2(s_g2_) {}
  snarkv_dk* handle() const {
    // lazy, once per key; the host mirror is multi-threaded (HostPool), so the check sits under a lock
    static std::mutex init_mu;
    std::lock_guard<std::mutex> init(init_mu);
    if (!dk_) {
      snarkv_dk* h = nullptr;
      DeviceScope lock;
      if (bn254_kzg_dk_create(svk.g.b, g2.b, s_g2.b, &h) != SNARKV_OK)
        throw std::runtime_error(std::string("bn254_kzg_dk_create: ") + snarkv_last_error());
      dk_ = std::shared_ptr<snarkv_dk>(h, [](snarkv_dk* p) { snarkv_dk_destroy(p); });
    }
    return dk_.get();
  }

 private:
  mutable std::shared_ptr<snarkv_dk> dk_;
};

// accumulation.rs:68-96
struct KzgAsProvingKey {
  std::optional<std::pair<G1Affine, G1Affine>> g;  // (g, s*g) when zero-knowledge
  bool zk() const { return g.has_value(); }
};
struct KzgAsVerifyingKey {
  bool zk_ = false;
  bool zk() const { return zk_; }
};
// accumulation.rs:98-137
struct KzgAsProof {
  std::optional<std::pair<G1Affine, G1Affine>> blind;
  Fr r;
  static Result<KzgAsProof> read(const KzgAsVerifyingKey& vk, const std::vector<KzgAccumulator>& instances,
                                 Transcript& t) {
    if (instances.empty()) throw Panic("KzgAsProof::read with no instances (reference: assert!, accumulation.rs:122)");
    for (auto& a : instances) {
      Error e = t.common_ec_point(a.lhs);
      if (!e.ok()) return Result<KzgAsProof>::Err(e);
      e = t.common_ec_point(a.rhs);
      if (!e.ok()) return Result<KzgAsProof>::Err(e);
    }
    KzgAsProof p;
    if (vk.zk()) {
      auto a = t.read_ec_point();
      if (!a.ok()) return Result<KzgAsProof>::Err(a.err);
      auto b = t.read_ec_point();
      if (!b.ok()) return Result<KzgAsProof>::Err(b.err);
      p.blind = std::make_pair(*a.value, *b.value);
    }
    p.r = t.squeeze_challenge();
    return Result<KzgAsProof>::Ok(p);
  }
};

struct Gwc19 {};
struct Bdfg21 {};

// ------------------------------------------------------------------ KzgAs
template <class MOS>
struct KzgAs {
  using Accumulator = KzgAccumulator;

  // accumulation.rs:30-39
  static Result<KzgAsProof> read_proof(const KzgAsVerifyingKey& vk, const std::vector<KzgAccumulator>& instances,
                                       Transcript& t) {
    return KzgAsProof::read(vk, instances, t);
  }

  // accumulation.rs:41-63: acc' = (sum r^i lhs_i, sum r^i rhs_i); blind pair last.
  // The two `evaluate(None)` go to the device as ONE segmented launch.
  static Result<KzgAccumulator> verify(const KzgAsVerifyingKey&, const std::vector<KzgAccumulator>& instances,
                                       const KzgAsProof& proof) {
    auto pairs = verify_pairs(instances, proof);
    std::vector<std::vector<std::pair<Fr, G1Affine>>> two;
    two.push_back(std::move(pairs.first));
    two.push_back(std::move(pairs.second));
    constexpr bool trace = SNARKV_HOST_TRACE != 0;  // dev aid: host / device split of this step on stderr
    auto t0 = std::chrono::steady_clock::now();
    auto pts = L::multi_scalar_multiplication_batch(two);
    if (trace)
      fprintf(stderr, "KzgAs::verify: %zu + %zu terms, device call %.3f ms\n", two[0].size(), two[1].size(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return Result<KzgAccumulator>::Ok(KzgAccumulator{pts[0], pts[1]});
  }
  // the host half of `verify`: the (scalar, base) lists of its two MSMs  sum r^i lhs_i ,  sum r^i rhs_i  (blind pair last),
  // for callers that put the MSMs of SEVERAL accumulation steps into one launch (aggregation.hpp)
  static std::pair<std::vector<std::pair<Fr, G1Affine>>, std::vector<std::pair<Fr, G1Affine>>> verify_pairs(
      const std::vector<KzgAccumulator>& instances, const KzgAsProof& proof) {
    std::vector<const G1Affine*> lhs, rhs;
    for (auto& a : instances) {
      lhs.push_back(&a.lhs);
      rhs.push_back(&a.rhs);
    }
    if (proof.blind) {
      lhs.push_back(&proof.blind->first);
      rhs.push_back(&proof.blind->second);
    }
    if (lhs.empty()) throw Panic("KzgAs::verify of no accumulators (reference: evaluate -> reduce().unwrap())");
    auto powers_of_r = proof.r.powers(lhs.size());
    std::vector<std::vector<std::pair<Fr, G1Affine>>> two;
    for (auto* bases : {&lhs, &rhs}) {
      // sum_i Msm::base(b_i) * r^i, as `push`es into ONE Msm: the same merge of equal bases in the same order as
      // `Msm::sum` of (m + 1) one-term Msms, without building them (0.5 ms of host time at 1 025 terms)
      MsmT m;
      for (size_t i = 0; i < bases->size(); ++i) m.push(powers_of_r[i], (*bases)[i]);
      two.push_back(m.pairs(std::nullopt));
    }
    return {std::move(two[0]), std::move(two[1])};
  }

  // accumulation.rs:148-197 (prover side; identical arithmetic).  `blind_scalar`
  // plays `M::Fr::random(rng)`.
  static Result<KzgAccumulator> create_proof(const KzgAsProvingKey& pk, const std::vector<KzgAccumulator>& instances,
                                             Transcript& t, const Fr& blind_scalar) {
    if (instances.empty()) throw Panic("create_proof with no instances (reference: assert!, accumulation.rs:159)");
    for (auto& a : instances) {
      Error e = t.common_ec_point(a.lhs);
      if (!e.ok()) return Result<KzgAccumulator>::Err(e);
      e = t.common_ec_point(a.rhs);
      if (!e.ok()) return Result<KzgAccumulator>::Err(e);
    }
    KzgAsProof proof;
    if (pk.zk()) {
      // lhs = s_g * s, rhs = g * s  (accumulation.rs:169-171): two 1-term MSMs
      auto pts = L::multi_scalar_multiplication_batch({{{blind_scalar, pk.g->second}}, {{blind_scalar, pk.g->first}}});
      Error e = t.write_ec_point(pts[0]);
      if (!e.ok()) return Result<KzgAccumulator>::Err(e);
      e = t.write_ec_point(pts[1]);
      if (!e.ok()) return Result<KzgAccumulator>::Err(e);
      proof.blind = std::make_pair(pts[0], pts[1]);
    }
    proof.r = t.squeeze_challenge();
    return verify(KzgAsVerifyingKey{pk.zk()}, instances, proof);
  }

  // decider.rs:70-82
  static Error decide(const KzgDecidingKey& dk, const KzgAccumulator& acc) {
    uint8_t a[128], ok = 0;
    acc.to_bytes(a);
    snarkv_dk* h = dk.handle();
    DeviceScope lock;
    int rc = bn254_kzg_dk_decide_batch(h, a, 1, &ok);
    if (rc < 0) throw std::runtime_error(std::string("bn254_kzg_dk_decide_batch: ") + snarkv_last_error());
    return ok ? Error{} : Error::assertion("e(lhs, g2)\xc2\xb7" "e(rhs, -s_g2) == O");
  }
  // decider.rs:84-93: ONE batched launch instead of a loop of pairings
  static Error decide_all(const KzgDecidingKey& dk, const std::vector<KzgAccumulator>& accs) {
    if (accs.empty()) return Error{};
    std::vector<uint8_t> a(128 * accs.size()), ok(accs.size());
    for (size_t i = 0; i < accs.size(); ++i) accs[i].to_bytes(&a[128 * i]);
    snarkv_dk* h = dk.handle();
    DeviceScope lock;
    int rc = bn254_kzg_dk_decide_batch(h, a.data(), accs.size(), ok.data());
    if (rc < 0) throw std::runtime_error(std::string("bn254_kzg_dk_decide_batch: ") + snarkv_last_error());
    for (uint8_t o : ok)
      if (!o) return Error::assertion("e(lhs, g2)\xc2\xb7" "e(rhs, -s_g2) == O");
    return Error{};
  }

  // PolynomialCommitmentScheme::verify -- specialised below per MOS
  template <class Proof>
  static Result<KzgAccumulator> pcs_verify(const KzgSuccinctVerifyingKey& svk, const std::vector<MsmT>& commitments,
                                           const Fr& z, const std::vector<Query<Fr>>& queries, const Proof& proof);
};

// ------------------------------------------------------------------ Gwc19
// gwc19.rs:84-110
struct Gwc19Proof {
  Fr v;
  std::vector<G1Affine> ws;
  Fr u;
};

namespace gwc19 {
struct QuerySet {
  Fr shift;
  std::vector<size_t> polys;
  std::vector<const Fr*> evals;
};
// gwc19.rs:142-160: group by shift, first-seen order
inline std::vector<QuerySet> query_sets(const std::vector<Query<Fr>>& queries) {
  std::vector<QuerySet> sets;
  for (auto& q : queries) {
    auto it = std::find_if(sets.begin(), sets.end(), [&](const QuerySet& s) { return s.shift == q.shift; });
    if (it != sets.end()) {
      it->polys.push_back(q.poly);
      it->evals.push_back(&q.eval);
    } else {
      sets.push_back(QuerySet{q.shift, {q.poly}, {&q.eval}});
    }
  }
  return sets;
}
// gwc19.rs:102-110
inline Result<Gwc19Proof> read(const std::vector<Query<std::monostate>>& queries, Transcript& t) {
  std::vector<Query<Fr>> qs;
  for (auto& q : queries) qs.push_back(Query<Fr>{q.poly, q.shift, Fr()});
  Gwc19Proof p;
  p.v = t.squeeze_challenge();
  auto ws = t.read_n_ec_points(query_sets(qs).size());
  if (!ws.ok()) return Result<Gwc19Proof>::Err(ws.err);
  p.ws = *ws.value;
  p.u = t.squeeze_challenge();
  return Result<Gwc19Proof>::Ok(p);
}
// gwc19.rs:45-82: the two Msm (lhs, rhs) before evaluation
inline std::pair<MsmT, MsmT> msms(const std::vector<MsmT>& commitments, const Fr& z,
                                  const std::vector<Query<Fr>>& queries, const Gwc19Proof& proof) {
  auto sets = query_sets(queries);
  auto powers_of_u = proof.u.powers(sets.size());
  size_t maxp = 0;
  for (auto& s : sets) maxp = std::max(maxp, s.polys.size());
  auto powers_of_v = proof.v.powers(maxp);
  std::vector<MsmT> per_set;
  for (size_t k = 0; k < sets.size(); ++k) {
    std::vector<MsmT> terms;  // QuerySet::msm, gwc19.rs:124-139
    for (size_t i = 0; i < sets[k].polys.size(); ++i)
      terms.push_back((commitments[sets[k].polys[i]] - MsmT::from_constant(*sets[k].evals[i])) * powers_of_v[i]);
    per_set.push_back(MsmT::sum(terms) * powers_of_u[k]);
  }
  MsmT f = MsmT::sum(per_set);
  std::vector<MsmT> rhs;
  for (size_t k = 0; k < proof.ws.size() && k < powers_of_u.size(); ++k)
    rhs.push_back(MsmT::base(&proof.ws[k]) * powers_of_u[k]);
  std::vector<MsmT> shifted;
  for (size_t k = 0; k < rhs.size(); ++k) shifted.push_back(rhs[k] * (L::load_const(sets[k].shift) * z));
  MsmT lhs = f + MsmT::sum(shifted);
  return {lhs, MsmT::sum(rhs)};
}
}  // namespace gwc19

template <>
template <>
inline Result<KzgAccumulator> KzgAs<Gwc19>::pcs_verify<Gwc19Proof>(const KzgSuccinctVerifyingKey& svk,
                                                                   const std::vector<MsmT>& commitments, const Fr& z,
                                                                   const std::vector<Query<Fr>>& queries,
                                                                   const Gwc19Proof& proof) {
  auto [lhs, rhs] = gwc19::msms(commitments, z, queries, proof);
  // gwc19.rs:79-80: the two `evaluate(Some(svk.g))` as one segmented launch
  auto pts = L::multi_scalar_multiplication_batch({lhs.pairs(svk.g), rhs.pairs(svk.g)});
  return Result<KzgAccumulator>::Ok(KzgAccumulator{pts[0], pts[1]});
}

#endif  // KZG

// ------------------------------------------------------------------ Bdfg21
// bdfg21.rs:85-120
struct Bdfg21Proof {
  Fr mu, gamma;
  G1Affine w;
  Fr z_prime;
  G1Affine w_prime;
  static Result<Bdfg21Proof> read(Transcript& t) {
    Bdfg21Proof p;
    p.mu = t.squeeze_challenge();
    p.gamma = t.squeeze_challenge();
    auto w = t.read_ec_point();
    if (!w.ok()) return Result<Bdfg21Proof>::Err(w.err);
    p.w = *w.value;
    p.z_prime = t.squeeze_challenge();
    auto wp = t.read_ec_point();
    if (!wp.ok()) return Result<Bdfg21Proof>::Err(wp.err);
    p.w_prime = *wp.value;
    return Result<Bdfg21Proof>::Ok(p);
  }
};

namespace bdfg21 {
struct QuerySet {
  std::vector<Fr> shifts;
  std::vector<size_t> polys;
  std::vector<std::vector<const Fr*>> evals;
};
inline bool contains(const std::vector<Fr>& v, const Fr& x) { return std::find(v.begin(), v.end(), x) != v.end(); }
inline bool same_set(const std::vector<Fr>& a, const std::vector<Fr>& b) {
  std::set<Fr> sa(a.begin(), a.end()), sb(b.begin(), b.end());
  return sa == sb;
}
// bdfg21.rs:121-171
inline std::vector<QuerySet> query_sets(const std::vector<Query<Fr>>& queries) {
  struct PS {
    size_t poly;
    std::vector<Fr> shifts;
    std::vector<const Fr*> evals;
  };
  std::vector<PS> poly_shifts;
  for (auto& q : queries) {
    auto it = std::find_if(poly_shifts.begin(), poly_shifts.end(), [&](const PS& p) { return p.poly == q.poly; });
    if (it != poly_shifts.end()) {
      if (!contains(it->shifts, q.shift)) {
        it->shifts.push_back(q.shift);
        it->evals.push_back(&q.eval);
      }
    } else {
      poly_shifts.push_back(PS{q.poly, {q.shift}, {&q.eval}});
    }
  }
  std::vector<QuerySet> sets;
  for (auto& ps : poly_shifts) {
    auto it = std::find_if(sets.begin(), sets.end(), [&](const QuerySet& s) { return same_set(s.shifts, ps.shifts); });
    if (it != sets.end()) {
      if (std::find(it->polys.begin(), it->polys.end(), ps.poly) == it->polys.end()) {
        it->polys.push_back(ps.poly);
        std::vector<const Fr*> ev;
        for (auto& lhs : it->shifts) {
          size_t idx = std::find(ps.shifts.begin(), ps.shifts.end(), lhs) - ps.shifts.begin();
          ev.push_back(ps.evals[idx]);
        }
        it->evals.push_back(ev);
      }
    } else {
      sets.push_back(QuerySet{ps.shifts, {ps.poly}, {ps.evals}});
    }
  }
  return sets;
}

// bdfg21.rs:246-371, after its two batch inversions (values, not Fractions)
struct QuerySetCoeff {
  Fr z_s;
  std::vector<Fr> eval_coeffs;            // barycentric weights
  std::optional<Fr> commitment_coeff;     // z_s_1 / z_s (None for the first set)
  Fr r_eval_coeff;
};

// bdfg21.rs:173-223
inline std::vector<QuerySetCoeff> query_set_coeffs(const std::vector<QuerySet>& sets, const Fr& z, const Fr& z_prime) {
  size_t size = 2;
  for (auto& s : sets) size = std::max(size, s.shifts.size());
  auto powers_of_z = z.powers(size);
  std::vector<QuerySetCoeff> coeffs;
  std::optional<Fr> z_s_1;
  // first batch inversion: barycentric-weight denominators and z_s (bdfg21.rs:218)
  for (auto& set : sets) {
    QuerySetCoeff c;
    const auto& shifts = set.shifts;
    const Fr& zk1 = powers_of_z[shifts.size() - 1];
    for (size_t j = 0; j < shifts.size(); ++j) {
      Fr ell = Fr::one();  // normalized_ell_prime, bdfg21.rs:269-281
      for (size_t i = 0; i < shifts.size(); ++i)
        if (i != j) ell *= (shifts[j] - shifts[i]);
      // sum_products_with_coeff: ell*z^(k-1)*z' - ell*shift*z^(k-1)*z   (bdfg21.rs:286-295)
      c.eval_coeffs.push_back(ell * zk1 * z_prime - ell * shifts[j] * zk1 * powers_of_z[1]);
    }
    c.z_s = Fr::one();
    for (auto& sh : shifts) c.z_s *= (z_prime - z * sh);  // bdfg21.rs:297-303
    if (z_s_1) c.commitment_coeff = c.z_s;                 // denominator for now
    else z_s_1 = c.z_s;
    coeffs.push_back(c);
  }
  {
    std::vector<Fr*> denoms;
    for (auto& c : coeffs) {
      for (auto& e : c.eval_coeffs) denoms.push_back(&e);
      if (c.commitment_coeff) denoms.push_back(&*c.commitment_coeff);
    }
    L::batch_invert(denoms);
  }
  for (auto& c : coeffs)
    if (c.commitment_coeff) *c.commitment_coeff = *z_s_1 * *c.commitment_coeff;
  // second batch inversion: the barycentric-weight sums (bdfg21.rs:219, :337-356)
  {
    std::vector<Fr*> denoms;
    for (auto& c : coeffs) {
      Fr sum;
      for (auto& e : c.eval_coeffs) sum += e;
      c.r_eval_coeff = sum;
      denoms.push_back(&c.r_eval_coeff);
    }
    L::batch_invert(denoms);
  }
  for (auto& c : coeffs)
    if (c.commitment_coeff) c.r_eval_coeff = *c.commitment_coeff * c.r_eval_coeff;
  return coeffs;
}

// bdfg21.rs:51-83
inline std::pair<MsmT, MsmT> msms(const std::vector<MsmT>& commitments, const Fr& z,
                                  const std::vector<Query<Fr>>& queries, const Bdfg21Proof& proof) {
  auto sets = query_sets(queries);
  auto coeffs = query_set_coeffs(sets, z, proof.z_prime);
  size_t maxp = 0;
  for (auto& s : sets) maxp = std::max(maxp, s.polys.size());
  auto powers_of_mu = proof.mu.powers(maxp);
  auto powers_of_gamma = proof.gamma.powers(sets.size());
  std::vector<MsmT> per_set;
  for (size_t k = 0; k < sets.size(); ++k) {
    const auto& set = sets[k];
    const auto& co = coeffs[k];
    std::vector<MsmT> terms;  // QuerySet::msm, bdfg21.rs:233-263
    for (size_t i = 0; i < set.polys.size(); ++i) {
      MsmT commitment = co.commitment_coeff ? commitments[set.polys[i]] * *co.commitment_coeff : commitments[set.polys[i]];
      Fr r_eval;
      for (size_t j = 0; j < co.eval_coeffs.size(); ++j) r_eval += co.eval_coeffs[j] * *set.evals[i][j];
      r_eval *= co.r_eval_coeff;
      terms.push_back((commitment - MsmT::from_constant(r_eval)) * powers_of_mu[i]);
    }
    per_set.push_back(MsmT::sum(terms) * powers_of_gamma[k]);
  }
  MsmT f = MsmT::sum(per_set) - MsmT::base(&proof.w) * coeffs[0].z_s;
  MsmT rhs = MsmT::base(&proof.w_prime);
  MsmT lhs = f + rhs * proof.z_prime;
  return {lhs, rhs};
}
}  // namespace bdfg21

#if !defined(SNARKV_HOST_PALLAS)
template <>
template <>
inline Result<KzgAccumulator> KzgAs<Bdfg21>::pcs_verify<Bdfg21Proof>(const KzgSuccinctVerifyingKey& svk,
                                                                     const std::vector<MsmT>& commitments, const Fr& z,
                                                                     const std::vector<Query<Fr>>& queries,
                                                                     const Bdfg21Proof& proof) {
  auto [lhs, rhs] = bdfg21::msms(commitments, z, queries, proof);
  // bdfg21.rs:80-81
  auto pts = L::multi_scalar_multiplication_batch({lhs.pairs(svk.g), rhs.pairs(svk.g)});
  return Result<KzgAccumulator>::Ok(KzgAccumulator{pts[0], pts[1]});
}

// --------------------------------------------------------- LimbsEncoding
// accumulator.rs:34-81 + util/arithmetic.rs:270-298 (LIMBS x BITS little-endian
// limbs of the four Fq coordinates, carried as Fr values).
template <size_t LIMBS, size_t BITS>
struct LimbsEncoding {
  static_assert(LIMBS * BITS <= 320, "");
  // fe_from_limbs: sum limb_i << (BITS i) as a canonical Fq (32 LE bytes); false if >= 2^256
  static bool fe_from_limbs(const Fr* limbs, uint8_t out[32]) {
    uint64_t acc[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < LIMBS; ++i) {
      uint8_t lb[32];
      limbs[i].to_bytes(lb);
      uint64_t lw[4];
      memcpy(lw, lb, 32);
      size_t shift = BITS * i, word = shift / 64, bit = shift % 64;
      uint64_t sh[5];
      sh[0] = lw[0] << bit;
      for (int k = 1; k < 4; ++k) sh[k] = (lw[k] << bit) | (bit ? lw[k - 1] >> (64 - bit) : 0);
      sh[4] = bit ? lw[3] >> (64 - bit) : 0;
      unsigned __int128 carry = 0;
      for (size_t k = 0; k < 5 || carry; ++k) {
        uint64_t add = k < 5 ? sh[k] : 0;
        if (word + k >= 6) {
          if (add || carry) return false;
          break;
        }
        unsigned __int128 t = (unsigned __int128)acc[word + k] + add + carry;
        acc[word + k] = (uint64_t)t;
        carry = t >> 64;
      }
    }
    if (acc[4] | acc[5]) return false;
    memcpy(out, acc, 32);
    return true;
  }
  // fe_to_limbs (arithmetic.rs:286-298)
  static void fe_to_limbs(const uint8_t fe[32], Fr* limbs) {
    for (size_t i = 0; i < LIMBS; ++i) {
      uint8_t lb[32] = {0};
      for (size_t bit = 0; bit < BITS; ++bit) {
        size_t src = BITS * i + bit;
        if (src >= 256) break;
        if ((fe[src / 8] >> (src % 8)) & 1) lb[bit / 8] |= (uint8_t)(1u << (bit % 8));
      }
      if (!Fr::from_bytes(lb, &limbs[i])) throw Panic("limb does not fit Fr");
    }
  }
  // accumulator.rs:57-81.  The reference PANICS on a non-canonical coordinate
  // (`from_repr().unwrap()`) or an off-curve point (`from_xy().unwrap()`).
  static Result<KzgAccumulator> from_repr(const std::vector<const Fr*>& limbs) {
    if (limbs.size() != 4 * LIMBS) throw Panic("LimbsEncoding::from_repr: wrong limb count (reference: assert_eq!)");
    uint8_t pts[128];
    for (size_t c = 0; c < 4; ++c) {
      Fr tmp[LIMBS];
      for (size_t i = 0; i < LIMBS; ++i) tmp[i] = *limbs[c * LIMBS + i];
      if (!fe_from_limbs(tmp, pts + 32 * c)) throw Panic("limbs overflow the base field (reference: from_repr().unwrap())");
    }
    // (0, 0) is this library's encoding of the identity and passes the device's on-curve check, but it is NOT
    // a curve point for `C::from_xy` (0 != 0 + 3): the reference's unwrap() panics on it
    static const uint8_t zero64[64] = {0};
    if (!memcmp(pts, zero64, 64) || !memcmp(pts + 64, zero64, 64))
      throw Panic("accumulator point (0, 0) is not on the curve (reference: from_xy().unwrap())");
    DeviceScope lock;
    if (bn254_g1_validate(pts, 2) != SNARKV_OK)
      throw Panic("accumulator point is non-canonical or off-curve (reference: from_xy().unwrap())");
    return Result<KzgAccumulator>::Ok(KzgAccumulator{G1Affine::from_bytes(pts), G1Affine::from_bytes(pts + 64)});
  }
  static std::vector<Fr> to_limbs(const KzgAccumulator& acc) {
    std::vector<Fr> out(4 * LIMBS);
    uint8_t b[128];
    acc.to_bytes(b);
    for (size_t c = 0; c < 4; ++c) fe_to_limbs(b + 32 * c, &out[c * LIMBS]);
    return out;
  }
};

#endif  // KZG

}  // namespace snarkv_host
