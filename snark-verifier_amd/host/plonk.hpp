// PLONK succinct verifier, front half: turns
// (protocol, instances, proof bytes) into the (scalar, base) lists the GPU
// path consumes, then decides.
//
//   reference                                                    here
//   `Rotation`, `Domain`      util/arithmetic.rs:98-160           Rotation (int32_t), Domain
//   `Query`, `Expression`,    verifier/plonk/protocol.rs:274-420  PQuery, Expression
//   `CommonPolynomial`
//   `PlonkProtocol`           protocol.rs:17-111                  PlonkProtocol
//   `CommonPolynomialEvaluation` protocol.rs:196-272              CommonPolyEval
//   `PlonkProof::{read, evaluations, commitments, queries}`       PlonkProof<MOS>
//                             verifier/plonk/proof.rs:52-349
//   `PlonkSuccinctVerifier`, `PlonkVerifier`  verifier/plonk.rs:58-147   same names
//
// Host Fr algebra only; every `evaluate` goes through GpuNativeLoader.  The
// batch form `verify_batch` collects the two MSMs of EVERY proof into one
// segmented launch (the data-parallel axis of SURVEY.md 8e).
#pragma once
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <type_traits>
#include <set>
#include <thread>

#include "ipa.hpp"
#include "pcs.hpp"

namespace snarkv_host {

// util/arithmetic.rs:123-160
struct Domain {
  size_t k = 0, n = 0;
  Fr n_inv, gen, gen_inv;
  static Domain make(size_t k, const Fr& gen) {
    Domain d;
    d.k = k;
    d.n = (size_t)1 << k;
    if (!Fr::from_u64((uint64_t)d.n).invert(&d.n_inv)) throw Panic("Domain: n not invertible");
    d.gen = gen;
    if (!gen.invert(&d.gen_inv)) throw Panic("Domain: generator not invertible (reference: unwrap, arithmetic.rs:141)");
    return d;
  }
  static Fr pow_u64(const Fr& b, uint64_t e) {  // `pow_vartime([e])`: only the significant bits
    Fr acc = Fr::one();
    for (int i = 63 - (e ? __builtin_clzll(e) : 63); i >= 0; --i) {
      acc = acc.square();
      if ((e >> i) & 1) acc = acc * b;
    }
    return acc;
  }
  Fr rotate_scalar(const Fr& scalar, int32_t rotation) const {  // arithmetic.rs:153-159
    if (rotation == 0) return scalar;
    if (rotation > 0) return scalar * pow_u64(gen, (uint64_t)rotation);
    return scalar * pow_u64(gen_inv, (uint64_t)(-(int64_t)rotation));
  }
};

// protocol.rs:287-301 (`Query { poly, rotation }`; named PQuery: pcs.hpp already has the PCS `Query`)
struct PQuery {
  size_t poly = 0;
  int32_t rotation = 0;
  bool operator<(const PQuery& o) const { return poly != o.poly ? poly < o.poly : rotation < o.rotation; }
  bool operator==(const PQuery& o) const { return poly == o.poly && rotation == o.rotation; }
};

// protocol.rs:303-315
struct Expression {
  enum Kind { Constant, Identity, Lagrange, Polynomial, Challenge, Negated, Sum, Product, Scaled, DistributePowers };
  Kind kind = Constant;
  Fr scalar;             // Constant, Scaled
  int32_t lagrange = 0;  // Lagrange(i)
  PQuery query;          // Polynomial
  size_t index = 0;      // Challenge
  std::vector<std::shared_ptr<Expression>> ch;  // operands; DistributePowers: exprs..., then the scalar expression
};
using ExprPtr = std::shared_ptr<Expression>;

struct InvalidProtocol : std::runtime_error {  // Error::InvalidProtocol(String), carried as an exception inside evaluate
  using std::runtime_error::runtime_error;
};

// `Expression::evaluate` (protocol.rs:317-373), same operand order
template <class T, class V>
T expr_evaluate(const Expression& e, V& v) {
  auto ev = [&](const ExprPtr& x) { return expr_evaluate<T, V>(*x, v); };
  switch (e.kind) {
    case Expression::Constant: return v.constant(e.scalar);
    case Expression::Identity: return v.common_identity();
    case Expression::Lagrange: return v.common_lagrange(e.lagrange);
    case Expression::Polynomial: return v.poly(e.query);
    case Expression::Challenge: return v.challenge(e.index);
    case Expression::Negated: return v.negated(ev(e.ch[0]));
    case Expression::Sum: {
      T a = ev(e.ch[0]);
      T b = ev(e.ch[1]);
      return v.sum(a, b);
    }
    case Expression::Product: {
      T a = ev(e.ch[0]);
      T b = ev(e.ch[1]);
      return v.product(a, b);
    }
    case Expression::Scaled: return v.scaled(ev(e.ch[0]), e.scalar);
    case Expression::DistributePowers: {
      if (e.ch.size() < 2) throw Panic("DistributePowers of no expressions (reference: assert!, protocol.rs:359)");
      size_t n = e.ch.size() - 1;
      if (n == 1) return ev(e.ch[0]);
      T acc = ev(e.ch[0]);
      T scalar = ev(e.ch[n]);
      for (size_t i = 1; i < n; ++i) acc = v.sum(v.product(acc, scalar), ev(e.ch[i]));
      return acc;
    }
  }
  throw Panic("bad expression kind");
}

// `used_langrange` / `used_query` (protocol.rs:391-420)
inline void expr_collect(const Expression& e, std::set<int32_t>* lagranges, std::set<PQuery>* queries) {
  if (e.kind == Expression::Lagrange && lagranges) lagranges->insert(e.lagrange);
  if (e.kind == Expression::Polynomial && queries) queries->insert(e.query);
  for (auto& c : e.ch) expr_collect(*c, lagranges, queries);
}

struct QuotientPolynomial {  // protocol.rs:274-285
  size_t chunk_degree = 1, num_chunk = 0;
  ExprPtr numerator;
};
struct InstanceCommittingKey {  // protocol.rs:541-547
  std::vector<G1Affine> bases;
  std::optional<G1Affine> constant;
};
enum class Linearization { None, WithoutConstant, MinusVanishingTimesQuotient };  // protocol.rs:528-539

// protocol.rs:17-72
struct PlonkProtocol {
  Domain domain;
  std::vector<G1Affine> preprocessed;
  std::vector<size_t> num_instance, num_witness, num_challenge;
  std::vector<PQuery> evaluations, queries;
  QuotientPolynomial quotient;
  std::optional<Fr> transcript_initial_state;
  std::optional<InstanceCommittingKey> instance_committing_key;
  Linearization linearization = Linearization::None;
  std::vector<std::vector<std::pair<size_t, size_t>>> accumulator_indices;

  // protocol.rs:80-111
  std::set<int32_t> langranges() const {
    std::set<int32_t> out;
    std::set<PQuery> used;
    expr_collect(*quotient.numerator, &out, &used);
    if (!instance_committing_key) {
      const size_t off = preprocessed.size();
      int32_t mn = 0, mx = 0;
      for (auto& q : used) {  // BTreeSet order
        if (q.poly < off || q.poly >= off + num_instance.size()) continue;
        if (q.rotation < mn) mn = q.rotation;
        else if (q.rotation > mx) mx = q.rotation;
      }
      size_t max_len = 0;
      for (size_t n : num_instance) max_len = std::max(max_len, n);
      for (int32_t i = -mx; i < (int32_t)max_len + (mn < 0 ? -mn : mn); ++i) out.insert(i);
    }
    return out;
  }
};

// protocol.rs:196-272; the fractions are evaluated on construction (the native
// loader's batch inversion, plonk.rs:66-70, here one inversion per denominator)
struct CommonPolyEval {
  Fr zn, zn_minus_one, zn_minus_one_inv, identity;
  std::map<int32_t, Fr> lagrange;
  CommonPolyEval(const Domain& domain, const std::set<int32_t>& langranges, const Fr& z) {
    zn = Domain::pow_u64(z, (uint64_t)domain.n);
    zn_minus_one = zn - Fr::one();
    Fr numer = zn_minus_one * domain.n_inv;
    identity = z;
    // denominators z - w^i and z^n - 1, inverted together (`L::batch_invert(denoms())`, plonk.rs:66-70):
    // one field inversion; zeros stay zero (loader.rs:255-262)
    std::vector<Fr> omegas, dens;
    for (int32_t i : langranges) {
      omegas.push_back(domain.rotate_scalar(Fr::one(), i));
      dens.push_back(z - omegas.back());
    }
    dens.push_back(zn_minus_one);
    std::vector<Fr> prefix(dens.size());
    Fr acc = Fr::one();
    for (size_t i = 0; i < dens.size(); ++i) {
      prefix[i] = acc;
      if (!dens[i].is_zero()) acc = acc * dens[i];
    }
    Fr inv;
    acc.invert(&inv);
    for (size_t i = dens.size(); i-- > 0;) {
      if (dens[i].is_zero()) continue;
      Fr d = dens[i];
      dens[i] = inv * prefix[i];
      inv = inv * d;
    }
    zn_minus_one_inv = dens.back();
    size_t k = 0;
    for (int32_t i : langranges) {
      lagrange[i] = numer * omegas[k] * dens[k];
      ++k;
    }
  }
  const Fr& get_lagrange(int32_t i) const {
    auto it = lagrange.find(i);
    if (it == lagrange.end()) throw Panic("missing Lagrange evaluation (reference: unwrap, protocol.rs:253)");
    return it->second;
  }
};

template <class MOS>
struct MosProof;
#if !defined(SNARKV_HOST_PALLAS)  // KZG: BN254 only
template <>
struct MosProof<Gwc19> {
  using type = Gwc19Proof;
};
template <>
struct MosProof<Bdfg21> {
  using type = Bdfg21Proof;
};
#endif
template <>
struct MosProof<Bgh19> {
  using type = Bgh19Proof;
};

// The `AS: AccumulationScheme + PolynomialCommitmentScheme` a multi-open scheme belongs to
// (plonk.rs:35-41): KzgAs<Gwc19 | Bdfg21> or IpaAs<Bgh19>.
// `impl AccumulatorEncoding for PhantomData<PCS>` (pcs.rs:173-184): the default encoding of schemes
// that define none (IPA) -- `from_repr` is `unimplemented!()`.
template <class Acc>
struct NoAccumulatorEncoding {
  static Result<Acc> from_repr(const std::vector<const Fr*>&) {
    throw Panic("AccumulatorEncoding::from_repr: unimplemented!() (reference pcs.rs:182)");
  }
};
#if !defined(SNARKV_HOST_PALLAS)  // KZG: BN254 only
template <class MOS>
struct PcsOf {
  using Svk = KzgSuccinctVerifyingKey;
  using Accumulator = KzgAccumulator;
  using DecidingKey = KzgDecidingKey;
  using DefaultAE = LimbsEncoding<4, 68>;
  static Error decide_all(const DecidingKey& dk, const std::vector<Accumulator>& accs) {
    return KzgAs<MOS>::decide_all(dk, accs);
  }
};
#else
template <class MOS>
struct PcsOf;
#endif
template <>
struct PcsOf<Bgh19> {
  using Svk = IpaSuccinctVerifyingKey;
  using Accumulator = IpaAccumulator;
  using DecidingKey = IpaDecidingKey;
  using DefaultAE = NoAccumulatorEncoding<IpaAccumulator>;
  static Error decide_all(const DecidingKey& dk, const std::vector<Accumulator>& accs) {
    return IpaAs<Bgh19>::decide_all(dk, accs);
  }
};

// proof.rs:18-45
template <class MOS>
struct PlonkProof {
  using PcsProof = typename MosProof<MOS>::type;
  std::optional<std::vector<G1Affine>> committed_instances;
  std::vector<G1Affine> witnesses;
  std::vector<Fr> challenges;
  std::vector<G1Affine> quotients;
  Fr z;
  std::vector<Fr> evaluations;
  PcsProof pcs;
  std::vector<typename PcsOf<MOS>::Accumulator> old_accumulators;

  // proof.rs:170-181
  static std::vector<Query<std::monostate>> empty_queries(const PlonkProtocol& pr) {
    std::vector<Query<std::monostate>> out;
    for (auto& q : pr.queries) out.push_back(Query<std::monostate>{q.poly, pr.domain.rotate_scalar(Fr::one(), q.rotation), {}});
    return out;
  }

  static Result<PcsProof> read_pcs(const typename PcsOf<MOS>::Svk& svk, const PlonkProtocol& pr, Transcript& t);

  // proof.rs:52-168; AE = LimbsEncoding<LIMBS, BITS>
  template <class AE = typename PcsOf<MOS>::DefaultAE>
  static Result<PlonkProof> read(const typename PcsOf<MOS>::Svk& svk, const PlonkProtocol& pr,
                                 const std::vector<std::vector<Fr>>& instances, Transcript& t) {
    using R = Result<PlonkProof>;
    if (pr.transcript_initial_state) {
      Error e = t.common_scalar(*pr.transcript_initial_state);
      if (!e.ok()) return R::Err(e);
    }
    if (pr.num_instance.size() != instances.size()) return R::Err(Error{Error::InvalidInstances, ""});
    for (size_t i = 0; i < instances.size(); ++i)
      if (pr.num_instance[i] != instances[i].size()) return R::Err(Error{Error::InvalidInstances, ""});
    PlonkProof p;
    if (pr.instance_committing_key) {
      const auto& ick = *pr.instance_committing_key;
      std::vector<std::vector<std::pair<Fr, G1Affine>>> jobs;
      for (auto& inst : instances) {
        std::vector<MsmT> terms;
        for (size_t i = 0; i < inst.size() && i < ick.bases.size(); ++i) terms.push_back(MsmT::base(&ick.bases[i]) * inst[i]);
        if (ick.constant) terms.push_back(MsmT::base(&*ick.constant));
        jobs.push_back(MsmT::sum(terms).pairs(std::nullopt));
      }
      // one segmented launch for all instance columns (each is an `evaluate(None)`, proof.rs:91-98)
      p.committed_instances = jobs.empty() ? std::vector<G1Affine>() : L::multi_scalar_multiplication_batch(jobs);
      for (auto& c : *p.committed_instances) {
        Error e = t.common_ec_point(c);
        if (!e.ok()) return R::Err(e);
      }
    } else {
      for (auto& inst : instances)
        for (auto& x : inst) {
          Error e = t.common_scalar(x);
          if (!e.ok()) return R::Err(e);
        }
    }
    const size_t phases = std::min(pr.num_witness.size(), pr.num_challenge.size());  // `zip`
    for (size_t ph = 0; ph < phases; ++ph) {
      auto w = t.read_n_ec_points(pr.num_witness[ph]);
      if (!w.ok()) return R::Err(w.err);
      p.witnesses.insert(p.witnesses.end(), w.value->begin(), w.value->end());
      auto c = t.squeeze_n_challenges(pr.num_challenge[ph]);
      p.challenges.insert(p.challenges.end(), c.begin(), c.end());
    }
    auto q = t.read_n_ec_points(pr.quotient.num_chunk);
    if (!q.ok()) return R::Err(q.err);
    p.quotients = *q.value;
    p.z = t.squeeze_challenge();
    for (size_t i = 0; i < pr.evaluations.size(); ++i) {
      auto s = t.read_scalar();
      if (!s.ok()) return R::Err(s.err);
      p.evaluations.push_back(*s.value);
    }
    auto pcs = read_pcs(svk, pr, t);
    if (!pcs.ok()) return R::Err(pcs.err);
    p.pcs = *pcs.value;
    for (auto& idx : pr.accumulator_indices) {
      std::vector<const Fr*> limbs;
      for (auto& ij : idx) {
        if (ij.first >= instances.size() || ij.second >= instances[ij.first].size())
          throw Panic("accumulator index out of range (reference: slice index panic, proof.rs:150)");
        limbs.push_back(&instances[ij.first][ij.second]);
      }
      auto acc = AE::from_repr(limbs);
      if (!acc.ok()) return R::Err(acc.err);
      p.old_accumulators.push_back(*acc.value);
    }
    return R::Ok(std::move(p));
  }

  // proof.rs:299-349
  std::map<PQuery, Fr> evaluations_map(const PlonkProtocol& pr, const std::vector<std::vector<Fr>>& instances,
                                       const CommonPolyEval& cpe) const {
    std::map<PQuery, Fr> evals;
    if (!pr.instance_committing_key) {
      const size_t off = pr.preprocessed.size();
      std::set<PQuery> used;
      expr_collect(*pr.quotient.numerator, nullptr, &used);
      for (auto& q : used) {
        if (q.poly < off || q.poly >= off + pr.num_instance.size()) continue;
        const auto& inst = instances[q.poly - off];
        Fr acc = Fr::zero();  // sum_products(instances, l_{i - r})
        for (size_t i = 0; i < inst.size(); ++i) acc = acc + inst[i] * cpe.get_lagrange((int32_t)i - q.rotation);
        evals[q] = acc;
      }
    }
    for (size_t i = 0; i < pr.evaluations.size() && i < evaluations.size(); ++i) evals[pr.evaluations[i]] = evaluations[i];
    return evals;
  }

  // proof.rs:199-297.  `storage` keeps the points the returned Msm borrow alive.
  std::vector<MsmT> commitments(const PlonkProtocol& pr, const CommonPolyEval& cpe, std::map<PQuery, Fr>& evals) const {
    std::vector<MsmT> cm;
    for (auto& p : pr.preprocessed) cm.push_back(MsmT::base(&p));
    if (committed_instances) {
      for (auto& p : *committed_instances) cm.push_back(MsmT::base(&p));
    } else {
      for (size_t i = 0; i < pr.num_instance.size(); ++i) cm.push_back(MsmT());
    }
    for (auto& p : witnesses) cm.push_back(MsmT::base(&p));

    struct V {
      const PlonkProof& self;
      const CommonPolyEval& cpe;
      std::map<PQuery, Fr>& evals;
      std::vector<MsmT>& cm;
      MsmT constant(const Fr& s) { return MsmT::from_constant(s); }
      MsmT common_identity() { return MsmT::from_constant(cpe.identity); }
      MsmT common_lagrange(int32_t i) { return MsmT::from_constant(cpe.get_lagrange(i)); }
      MsmT poly(const PQuery& q) {
        auto it = evals.find(q);
        if (it != evals.end()) return MsmT::from_constant(it->second);
        if (q.rotation == 0 && q.poly < cm.size()) return cm[q.poly];
        throw InvalidProtocol("Missing query");
      }
      MsmT challenge(size_t i) {
        if (i >= self.challenges.size()) throw InvalidProtocol("Missing challenge");
        return MsmT::from_constant(self.challenges[i]);
      }
      MsmT negated(const MsmT& a) { return -a; }
      MsmT sum(const MsmT& a, const MsmT& b) { return a + b; }
      MsmT product(const MsmT& a, const MsmT& b) {
        if (a.size() == 0) return b * *a.try_into_constant();
        if (b.size() == 0) return a * *b.try_into_constant();
        throw InvalidProtocol("Invalid linearization");
      }
      MsmT scaled(const MsmT& a, const Fr& s) { return a * s; }
    } v{*this, cpe, evals, cm};
    // When every polynomial the numerator touches has an evaluation (always so
    // without linearization) the whole tree is scalar arithmetic: evaluate it on
    // Fr and wrap the result, instead of building an Msm object per node.
    struct NeedMsm {};
    struct VF {
      const PlonkProof& self;
      const CommonPolyEval& cpe;
      std::map<PQuery, Fr>& evals;
      Fr constant(const Fr& s) { return s; }
      Fr common_identity() { return cpe.identity; }
      Fr common_lagrange(int32_t i) { return cpe.get_lagrange(i); }
      Fr poly(const PQuery& q) {
        auto it = evals.find(q);
        if (it == evals.end()) throw NeedMsm{};
        return it->second;
      }
      Fr challenge(size_t i) {
        if (i >= self.challenges.size()) throw InvalidProtocol("Missing challenge");
        return self.challenges[i];
      }
      Fr negated(const Fr& a) { return -a; }
      Fr sum(const Fr& a, const Fr& b) { return a + b; }
      Fr product(const Fr& a, const Fr& b) { return a * b; }
      Fr scaled(const Fr& a, const Fr& s) { return a * s; }
    } vf{*this, cpe, evals};
    MsmT numerator;
    try {
      numerator = MsmT::from_constant(expr_evaluate<Fr>(*pr.quotient.numerator, vf));
    } catch (const NeedMsm&) {
      numerator = expr_evaluate<MsmT>(*pr.quotient.numerator, v);
    }

    PQuery quotient_query{pr.preprocessed.size() + pr.num_instance.size() + witnesses.size(), 0};
    auto coeffs = Domain::pow_u64(cpe.zn, (uint64_t)pr.quotient.chunk_degree).powers(quotients.size());
    std::vector<MsmT> chunks;
    for (size_t i = 0; i < quotients.size(); ++i) chunks.push_back(MsmT::base(&quotients[i]) * coeffs[i]);
    MsmT quotient = MsmT::sum(chunks);
    switch (pr.linearization) {
      case Linearization::WithoutConstant: {
        PQuery lq{quotient_query.poly + 1, 0};
        auto [msm, constant] = numerator.split();
        cm.push_back(quotient);
        cm.push_back(msm);
        auto it = evals.find(lq);
        if (it == evals.end()) throw Panic("missing linearization evaluation (reference: unwrap, proof.rs:267)");
        evals[quotient_query] = ((constant ? *constant : Fr::zero()) + it->second) * cpe.zn_minus_one_inv;
        break;
      }
      case Linearization::MinusVanishingTimesQuotient: {
        auto [msm, constant] = (numerator - quotient * cpe.zn_minus_one).split();
        cm.push_back(msm);
        evals[quotient_query] = constant ? *constant : Fr::zero();
        break;
      }
      case Linearization::None: {
        cm.push_back(quotient);
        if (numerator.size() != 0) throw InvalidProtocol("Invalid linearization");
        evals[quotient_query] = *numerator.try_into_constant() * cpe.zn_minus_one_inv;
        break;
      }
    }
    return cm;
  }

  // proof.rs:183-197
  std::vector<Query<Fr>> queries(const PlonkProtocol& pr, const std::map<PQuery, Fr>& evals) const {
    std::vector<Query<Fr>> out;
    auto eq = empty_queries(pr);
    for (size_t i = 0; i < eq.size(); ++i) {
      auto it = evals.find(pr.queries[i]);
      if (it == evals.end()) throw Panic("query without evaluation (reference: unwrap, proof.rs:193)");
      out.push_back(Query<Fr>{eq[i].poly, eq[i].shift, it->second});
    }
    return out;
  }
};

#if !defined(SNARKV_HOST_PALLAS)  // KZG: BN254 only
template <>
inline Result<Gwc19Proof> PlonkProof<Gwc19>::read_pcs(const KzgSuccinctVerifyingKey&, const PlonkProtocol& pr,
                                                      Transcript& t) {
  return gwc19::read(empty_queries(pr), t);
}
template <>
inline Result<Bdfg21Proof> PlonkProof<Bdfg21>::read_pcs(const KzgSuccinctVerifyingKey&, const PlonkProtocol&,
                                                        Transcript& t) {
  return Bdfg21Proof::read(t);
}
#endif
template <>
inline Result<Bgh19Proof> PlonkProof<Bgh19>::read_pcs(const IpaSuccinctVerifyingKey& svk, const PlonkProtocol& pr,
                                                      Transcript& t) {
  std::vector<Query<Fr>> qs;  // only (poly, shift) matter to `Bgh19Proof::read` (the number of rotation sets)
  for (auto& q : empty_queries(pr)) qs.push_back(Query<Fr>{q.poly, q.shift, Fr()});
  return IpaBgh19::read_proof(svk, qs, t);
}

#if !defined(SNARKV_HOST_PALLAS)  // KZG: BN254 only
namespace plonk_detail {
inline std::pair<MsmT, MsmT> pcs_msms(const std::vector<MsmT>& cm, const Fr& z, const std::vector<Query<Fr>>& q,
                                      const Gwc19Proof& p) {
  return gwc19::msms(cm, z, q, p);
}
inline std::pair<MsmT, MsmT> pcs_msms(const std::vector<MsmT>& cm, const Fr& z, const std::vector<Query<Fr>>& q,
                                      const Bdfg21Proof& p) {
  return bdfg21::msms(cm, z, q, p);
}
}  // namespace plonk_detail
#endif

// cost.rs:5-33 + the `CostEstimation` impls (gwc19.rs:168-175, bdfg21.rs:379-385, plonk.rs:149-188)
struct Cost {
  size_t num_instance = 0, num_commitment = 0, num_evaluation = 0, num_msm = 0, num_pairing = 0;
  Cost operator+(const Cost& o) const {
    return Cost{num_instance + o.num_instance, num_commitment + o.num_commitment, num_evaluation + o.num_evaluation,
                num_msm + o.num_msm, num_pairing + o.num_pairing};
  }
  bool operator==(const Cost& o) const {
    return num_instance == o.num_instance && num_commitment == o.num_commitment && num_evaluation == o.num_evaluation &&
           num_msm == o.num_msm && num_pairing == o.num_pairing;
  }
};
#if !defined(SNARKV_HOST_PALLAS)  // KZG: BN254 only
inline Cost pcs_estimate_cost(Gwc19, const std::vector<Query<std::monostate>>& queries) {
  std::vector<Query<Fr>> qs;
  for (auto& q : queries) qs.push_back(Query<Fr>{q.poly, q.shift, Fr()});
  size_t num_w = gwc19::query_sets(qs).size();
  Cost c;
  c.num_commitment = num_w;
  c.num_msm = num_w;
  return c;
}
inline Cost pcs_estimate_cost(Bdfg21, const std::vector<Query<std::monostate>>&) {
  Cost c;
  c.num_commitment = 2;
  c.num_msm = 2;
  return c;
}
#endif

// verifier/plonk.rs:32-92
template <class MOS>
struct PlonkSuccinctVerifier {
  using Proof = PlonkProof<MOS>;
  using Pairs = std::vector<std::pair<Fr, G1Affine>>;
  using Svk = typename PcsOf<MOS>::Svk;
  using Accumulator = typename PcsOf<MOS>::Accumulator;

  static Result<Proof> read_proof(const Svk& svk, const PlonkProtocol& pr, const std::vector<std::vector<Fr>>& instances,
                                  Transcript& t) {
    return Proof::read(svk, pr, instances, t);
  }

#if !defined(SNARKV_HOST_PALLAS)  // KZG: BN254 only
  // the host part of `verify`: the (lhs, rhs) pair lists of the PCS accumulator
  static Result<std::pair<Pairs, Pairs>> msm_pairs(const KzgSuccinctVerifyingKey& svk, const PlonkProtocol& pr,
                                                   const std::vector<std::vector<Fr>>& instances, const Proof& proof) {
    using R = Result<std::pair<Pairs, Pairs>>;
    try {
      CommonPolyEval cpe(pr.domain, pr.langranges(), proof.z);
      auto evals = proof.evaluations_map(pr, instances, cpe);
      auto cm = proof.commitments(pr, cpe, evals);
      auto queries = proof.queries(pr, evals);
      auto [lhs, rhs] = plonk_detail::pcs_msms(cm, proof.z, queries, proof.pcs);
      return R::Ok({lhs.pairs(svk.g), rhs.pairs(svk.g)});
    } catch (const InvalidProtocol& e) {
      return R::Err(Error{Error::InvalidProtocol, e.what()});
    }
  }

#endif
  // plonk.rs:58-92: [new accumulator] ++ old accumulators
  static Result<std::vector<Accumulator>> verify(const Svk& svk, const PlonkProtocol& pr,
                                                 const std::vector<std::vector<Fr>>& instances, const Proof& proof) {
    using R = Result<std::vector<Accumulator>>;
    std::vector<Accumulator> out;
    if constexpr (std::is_same_v<MOS, Bgh19>) {
      // `AS::verify` of IpaAs<Bgh19> (bgh19.rs:48-96): the succinct check is part of it
      try {
        CommonPolyEval cpe(pr.domain, pr.langranges(), proof.z);
        auto evals = proof.evaluations_map(pr, instances, cpe);
        auto cm = proof.commitments(pr, cpe, evals);
        auto queries = proof.queries(pr, evals);
        auto acc = IpaBgh19::verify(svk, cm, proof.z, queries, proof.pcs);
        if (!acc.ok()) return R::Err(acc.err);
        out.push_back(std::move(*acc.value));
      } catch (const InvalidProtocol& e) {
        return R::Err(Error{Error::InvalidProtocol, e.what()});
      }
    }
#if !defined(SNARKV_HOST_PALLAS)
    else {
      auto prs = msm_pairs(svk, pr, instances, proof);
      if (!prs.ok()) return R::Err(prs.err);
      auto pts = L::multi_scalar_multiplication_batch({prs.value->first, prs.value->second});
      out.push_back(KzgAccumulator{pts[0], pts[1]});
    }
#endif
    out.insert(out.end(), proof.old_accumulators.begin(), proof.old_accumulators.end());
    return R::Ok(out);
  }

#if !defined(SNARKV_HOST_PALLAS)  // KZG: BN254 only
  // plonk.rs:149-176
  static Cost estimate_cost(const PlonkProtocol& pr) {
    Cost c;
    size_t nw = 0;
    for (size_t x : pr.num_witness) nw += x;
    for (size_t x : pr.num_instance) c.num_instance += x;
    c.num_commitment = nw + pr.quotient.num_chunk;
    c.num_evaluation = pr.evaluations.size();
    c.num_msm = pr.preprocessed.size() + c.num_commitment + 1 + 2 * pr.accumulator_indices.size();
    return c + pcs_estimate_cost(MOS{}, Proof::empty_queries(pr));
  }

  // Many proofs (possibly of different protocols): all 2 x N MSMs in ONE segmented launch.
  static Result<std::vector<std::vector<KzgAccumulator>>> verify_batch(
      const KzgSuccinctVerifyingKey& svk, const std::vector<const PlonkProtocol*>& protocols,
      const std::vector<std::vector<std::vector<Fr>>>& instances, const std::vector<Proof>& proofs,
      unsigned threads = 1) {
    using R = Result<std::vector<std::vector<KzgAccumulator>>>;
    std::vector<Pairs> jobs(2 * proofs.size());
    std::vector<Error> errs(proofs.size());
    parallel_for(proofs.size(), threads, [&](size_t i) {
      auto prs = msm_pairs(svk, *protocols[i], instances[i], proofs[i]);
      if (!prs.ok()) {
        errs[i] = prs.err;
        return;
      }
      jobs[2 * i] = std::move(prs.value->first);
      jobs[2 * i + 1] = std::move(prs.value->second);
    }, 2);
    for (auto& e : errs)
      if (!e.ok()) return R::Err(e);
    auto pts = jobs.empty() ? std::vector<G1Affine>() : L::multi_scalar_multiplication_batch(jobs);
    std::vector<std::vector<KzgAccumulator>> out;
    for (size_t i = 0; i < proofs.size(); ++i) {
      std::vector<KzgAccumulator> a{KzgAccumulator{pts[2 * i], pts[2 * i + 1]}};
      a.insert(a.end(), proofs[i].old_accumulators.begin(), proofs[i].old_accumulators.end());
      out.push_back(std::move(a));
    }
    return R::Ok(out);
  }
#endif
};

// Many PLONK-over-IPA proofs (possibly of different protocols): the host halves in parallel, then all
// 2 x N MSMs of the succinct checks in ONE segmented launch -- the IPA counterpart of
// PlonkSuccinctVerifier<Gwc19 | Bdfg21>::verify_batch.  One accumulator per proof, in order.
inline Result<std::vector<IpaAccumulator>> plonk_ipa_verify_batch(
    const IpaSuccinctVerifyingKey& svk, const std::vector<const PlonkProtocol*>& protocols,
    const std::vector<std::vector<std::vector<Fr>>>& instances, const std::vector<PlonkProof<Bgh19>>& proofs,
    unsigned threads = 1) {
  using R = Result<std::vector<IpaAccumulator>>;
  std::vector<Ipa::Pending> pending(proofs.size());
  std::vector<Error> errs(proofs.size());
  parallel_for(proofs.size(), threads, [&](size_t i) {
    try {
      const PlonkProtocol& pr = *protocols[i];
      CommonPolyEval cpe(pr.domain, pr.langranges(), proofs[i].z);
      auto evals = proofs[i].evaluations_map(pr, instances[i], cpe);
      auto cm = proofs[i].commitments(pr, cpe, evals);
      auto queries = proofs[i].queries(pr, evals);
      pending[i] = IpaBgh19::verify_pairs(svk, cm, proofs[i].z, queries, proofs[i].pcs);
    } catch (const InvalidProtocol& e) {
      errs[i] = Error{Error::InvalidProtocol, e.what()};
    }
  });
  for (auto& e : errs)
    if (!e.ok()) return R::Err(e);
  return Ipa::finish_batch(pending);
}

// verifier/plonk.rs:94-147: succinct verify, then `decide_all`
template <class MOS>
struct PlonkVerifier {
  using Proof = PlonkProof<MOS>;
  using DecidingKey = typename PcsOf<MOS>::DecidingKey;
  static Result<Proof> read_proof(const DecidingKey& vk, const PlonkProtocol& pr,
                                  const std::vector<std::vector<Fr>>& instances, Transcript& t) {
    return Proof::read(vk.svk, pr, instances, t);
  }
  static Error verify(const DecidingKey& vk, const PlonkProtocol& pr, const std::vector<std::vector<Fr>>& instances,
                      const Proof& proof) {
    auto accs = PlonkSuccinctVerifier<MOS>::verify(vk.svk, pr, instances, proof);
    if (!accs.ok()) return accs.err;
    return PcsOf<MOS>::decide_all(vk, *accs.value);
  }
#if !defined(SNARKV_HOST_PALLAS)  // KZG: BN254 only
  // plonk.rs:178-188
  static Cost estimate_cost(const PlonkProtocol& pr) {
    Cost c = PlonkSuccinctVerifier<MOS>::estimate_cost(pr);
    c.num_pairing += 2;
    return c;
  }
#endif
};

}  // namespace snarkv_host
