// Mirror of the reference's inner-product-argument layer on the native loader:
//   h_eval / h_coeffs                        snark-verifier/src/pcs/ipa.rs:391-421
//   IpaSuccinctVerifyingKey                  snark-verifier/src/pcs/ipa.rs:251-276
//   IpaProof::read, xi, xi_inv               snark-verifier/src/pcs/ipa.rs:278-374
//   Ipa::succinct_verify                     snark-verifier/src/pcs/ipa.rs:139-180
//   IpaAccumulator                           snark-verifier/src/pcs/ipa/accumulator.rs:3-25
//   IpaAs::{read_proof, verify}, IpaAsProof  snark-verifier/src/pcs/ipa/accumulation.rs:21-146
//   IpaDecidingKey, decide / decide_all      snark-verifier/src/pcs/ipa/decider.rs:3-68
//   Bgh19 (halo2's IPA multi-open)           snark-verifier/src/pcs/ipa/multiopen/bgh19.rs:21-399
//
// `IpaAs::decide` is the reference's second consumer of the large-MSM hot path:
// U == `util::msm::multi_scalar_multiplication(h_coeffs(xi, 1), dk.g)` with 2^k
// terms (decider.rs:51-52) -- here ONE device Pippenger
// over the device-resident committing key (`bn254_ipa_decide_batch`: h_coeffs is
// built by a kernel, so a decide moves k scalars in and 64 bytes out); the two
// `evaluate(None)` of the succinct check (ipa.rs:172,177) go to the device as one
// segmented launch.  The other scalars (h_eval, one batch inversion) stay on the
// host, as in the reference.
//
// The scheme is generic over `C: CurveAffine`; the reference's tests use pallas,
// this mirror BN254 G1 -- the curve the device kernels are built for.  The prover
// halves (`Ipa::create_proof`, `IpaAs::create_proof`) live in oracle/ipa.py only:
// they make the test proofs and are not part of the verification path.
#pragma once
#include <memory>
#include <optional>
#include <vector>

#include "pcs.hpp"

namespace snarkv_host {

// ipa.rs:391-403: prod_i (z^(2^i) xi_{k-1-i} + 1)
inline Fr h_eval(const std::vector<Fr>& xi, const Fr& z) {
  Fr out = Fr::one(), zp = z;
  for (size_t i = xi.size(); i-- > 0;) {
    out *= zp * xi[i] + Fr::one();
    zp = zp * zp;
  }
  return out;
}

// ipa.rs:405-421
inline std::vector<Fr> h_coeffs(const std::vector<Fr>& xi, const Fr& scalar) {
  if (xi.empty()) throw Panic("h_coeffs of no challenges (reference: assert!, ipa.rs:406)");
  std::vector<Fr> coeffs((size_t)1 << xi.size());
  coeffs[0] = scalar;
  size_t len = 1;
  for (size_t i = xi.size(); i-- > 0; len <<= 1) {
    const Fr& x = xi[i];
    if (len >= 4096) {
      parallel_for(len / 1024, 16, [&](size_t t) {
        for (size_t j = t * 1024; j < (t + 1) * 1024; ++j) coeffs[len + j] = coeffs[j] * x;
      }, 1);
    } else {
      for (size_t j = 0; j < len; ++j) coeffs[len + j] = coeffs[j] * x;
    }
  }
  return coeffs;
}

// ipa.rs:251-276 (`domain` enters only through k)
struct IpaSuccinctVerifyingKey {
  size_t k = 0;
  G1Affine g;  // G_0
  G1Affine h;
  std::optional<G1Affine> s;
  bool zk() const { return s.has_value(); }
};

// accumulator.rs:3-25
struct IpaAccumulator {
  std::vector<Fr> xi;
  G1Affine u;
};

// ipa.rs:278-389
struct IpaProof {
  struct Round {
    G1Affine l, r;
    Fr xi;
  };
  std::optional<std::pair<G1Affine, Fr>> c_bar_alpha;
  std::optional<Fr> omega_prime;
  Fr xi_0;
  std::vector<Round> rounds;
  G1Affine u;
  Fr c;

  // ipa.rs:320-356
  static Result<IpaProof> read(const IpaSuccinctVerifyingKey& svk, Transcript& t) {
    using Res = Result<IpaProof>;
    IpaProof p;
    if (svk.zk()) {
      auto c_bar = t.read_ec_point();
      if (!c_bar.ok()) return Res::Err(c_bar.err);
      Fr alpha = t.squeeze_challenge();
      p.c_bar_alpha = std::make_pair(*c_bar.value, alpha);
      auto om = t.read_scalar();
      if (!om.ok()) return Res::Err(om.err);
      p.omega_prime = *om.value;
    }
    p.xi_0 = t.squeeze_challenge();
    for (size_t i = 0; i < svk.k; ++i) {
      auto l = t.read_ec_point();
      if (!l.ok()) return Res::Err(l.err);
      auto r = t.read_ec_point();
      if (!r.ok()) return Res::Err(r.err);
      p.rounds.push_back(Round{*l.value, *r.value, t.squeeze_challenge()});
    }
    auto u = t.read_ec_point();
    if (!u.ok()) return Res::Err(u.err);
    auto c = t.read_scalar();
    if (!c.ok()) return Res::Err(c.err);
    p.u = *u.value;
    p.c = *c.value;
    return Res::Ok(std::move(p));
  }

  // ipa.rs:358-361
  std::vector<Fr> xi() const {
    std::vector<Fr> v;
    for (auto& r : rounds) v.push_back(r.xi);
    return v;
  }
  // ipa.rs:363-373: `Fraction::one_over` + the loader's batch_invert (zeros stay zero)
  std::vector<Fr> xi_inv() const {
    std::vector<Fr> v = xi();
    std::vector<Fr*> ptrs;
    for (auto& x : v) ptrs.push_back(&x);
    L::batch_invert(ptrs);
    return v;
  }
};

struct Ipa {
  // ipa.rs:127-136
  static Result<IpaProof> read_proof(const IpaSuccinctVerifyingKey& svk, Transcript& t) {
    return IpaProof::read(svk, t);
  }

  // The host half of `succinct_verify` (ipa.rs:139-180): the pair lists of
  //   lhs = C' + eval [H'] + sum (xi_i^-1 [L_i] + xi_i [R_i])   and   rhs = c [U] + v' [H'],
  // plus the accumulator that results if they turn out equal.  No device work.
  struct Pending {
    std::vector<std::pair<Fr, G1Affine>> lhs, rhs;
    IpaAccumulator acc;
  };
  static Pending succinct_verify_pairs(const IpaSuccinctVerifyingKey& svk, const MsmT& commitment, const Fr& z,
                                       const Fr& eval, const IpaProof& proof) {
    const G1Affine h = L::ec_point_load_const(svk.h);
    MsmT h_prime = MsmT::base(&h) * proof.xi_0;
    MsmT c_prime = commitment;
    if (svk.zk() && proof.c_bar_alpha && proof.omega_prime) {
      c_prime += MsmT::base(&proof.c_bar_alpha->first) * proof.c_bar_alpha->second;
      c_prime -= MsmT::base(&*svk.s) * *proof.omega_prime;
    } else if (svk.zk() || proof.c_bar_alpha || proof.omega_prime) {
      throw Panic("IPA proof / key disagree on zero-knowledge (reference: unreachable!, ipa.rs:160)");
    }
    MsmT c_k = c_prime + h_prime * eval;
    const std::vector<Fr> xi = proof.xi(), xi_inv = proof.xi_inv();
    for (size_t i = 0; i < proof.rounds.size(); ++i) {
      c_k += MsmT::base(&proof.rounds[i].l) * xi_inv[i];
      c_k += MsmT::base(&proof.rounds[i].r) * xi[i];
    }
    Fr v_prime = h_eval(xi, z) * proof.c;
    MsmT rhs = MsmT::base(&proof.u) * proof.c + h_prime * v_prime;
    return Pending{c_k.pairs(std::nullopt), rhs.pairs(std::nullopt), IpaAccumulator{xi, proof.u}};  // pairs hold copies
  }

  // ipa.rs:139-180: the two `evaluate(None)` as one segmented launch, then `ec_point_assert_eq`
  static Result<IpaAccumulator> succinct_verify(const IpaSuccinctVerifyingKey& svk, const MsmT& commitment, const Fr& z,
                                                const Fr& eval, const IpaProof& proof) {
    Pending p = succinct_verify_pairs(svk, commitment, z, eval, proof);
    auto pts = L::multi_scalar_multiplication_batch({p.lhs, p.rhs});
    Error e = L::ec_point_assert_eq("C_k == c[U] + v'[H']", pts[0], pts[1]);
    if (!e.ok()) return Result<IpaAccumulator>::Err(e);
    return Result<IpaAccumulator>::Ok(std::move(p.acc));
  }

  // Many openings at once: every (lhs, rhs) of every opening in ONE segmented launch.
  static Result<std::vector<IpaAccumulator>> finish_batch(std::vector<Pending>& pending) {
    using R = Result<std::vector<IpaAccumulator>>;
    std::vector<std::vector<std::pair<Fr, G1Affine>>> jobs;
    for (auto& p : pending) {
      jobs.push_back(std::move(p.lhs));
      jobs.push_back(std::move(p.rhs));
    }
    auto pts = jobs.empty() ? std::vector<G1Affine>() : L::multi_scalar_multiplication_batch(jobs);
    std::vector<IpaAccumulator> out;
    for (size_t i = 0; i < pending.size(); ++i) {
      Error e = L::ec_point_assert_eq("C_k == c[U] + v'[H']", pts[2 * i], pts[2 * i + 1]);
      if (!e.ok()) return R::Err(e);
      out.push_back(std::move(pending[i].acc));
    }
    return R::Ok(std::move(out));
  }
};

// accumulation.rs:81-146
struct IpaAsProof {
  struct Abu {
    Fr a, b;
    G1Affine u;
  };
  std::optional<Abu> a_b_u;
  std::optional<Fr> omega;
  Fr alpha, z;
  IpaProof ipa;

  static Result<IpaAsProof> read(const IpaSuccinctVerifyingKey& vk, const std::vector<IpaAccumulator>& instances,
                                 Transcript& t) {
    using Res = Result<IpaAsProof>;
    if (instances.size() <= 1) throw Panic("IpaAsProof::read needs > 1 instances (reference: assert!, accumulation.rs:107)");
    IpaAsProof p;
    if (vk.zk()) {
      auto a = t.read_scalar();
      if (!a.ok()) return Res::Err(a.err);
      auto b = t.read_scalar();
      if (!b.ok()) return Res::Err(b.err);
      auto u = t.read_ec_point();
      if (!u.ok()) return Res::Err(u.err);
      p.a_b_u = Abu{*a.value, *b.value, *u.value};
      auto om = t.read_scalar();
      if (!om.ok()) return Res::Err(om.err);
      p.omega = *om.value;
    }
    for (auto& acc : instances) {
      for (auto& x : acc.xi) {
        Error e = t.common_scalar(x);
        if (!e.ok()) return Res::Err(e);
      }
      Error e = t.common_ec_point(acc.u);
      if (!e.ok()) return Res::Err(e);
    }
    p.alpha = t.squeeze_challenge();
    p.z = t.squeeze_challenge();
    auto ipa = IpaProof::read(vk, t);
    if (!ipa.ok()) return Res::Err(ipa.err);
    p.ipa = std::move(*ipa.value);
    return Res::Ok(std::move(p));
  }
};

// decider.rs:3-22.  The committing key is uploaded to the device once and cached (the
// reference walks `dk.g` in host memory on every decide).
struct IpaDecidingKey {
  IpaSuccinctVerifyingKey svk;
  std::vector<G1Affine> g;
  snarkv_ipa_dk* handle() const {
    static std::mutex init_mu;  // lazy, once per key, under a lock (the host mirror is multi-threaded)
    std::lock_guard<std::mutex> init(init_mu);
    if (!dk_) {
      if (g.empty() || (g.size() & (g.size() - 1)) != 0 || g.size() != ((size_t)1 << svk.k))
        throw Panic("IpaDecidingKey: g must hold 2^k points (reference: assert_eq!(scalars.len(), bases.len()), msm.rs:309)");
      static_assert(sizeof(G1Affine) == 64, "G1Affine is the 64-byte wire form");
      snarkv_ipa_dk* h = nullptr;
      DeviceScope lock;
      if (SNARKV_DEV(ipa_dk_create)(g[0].b, g.size(), &h) != SNARKV_OK)
        throw std::runtime_error(std::string("ipa_dk_create: ") + SNARKV_DEV_LAST_ERROR());
      dk_ = std::shared_ptr<snarkv_ipa_dk>(h, [](snarkv_ipa_dk* p) { SNARKV_DEV_IPA_DK_DESTROY(p); });
    }
    return dk_.get();
  }

 private:
  mutable std::shared_ptr<snarkv_ipa_dk> dk_;
};

template <class MOS = std::monostate>
struct IpaAs {
  using Accumulator = IpaAccumulator;

  // accumulation.rs:34-43
  static Result<IpaAsProof> read_proof(const IpaSuccinctVerifyingKey& vk, const std::vector<IpaAccumulator>& instances,
                                       Transcript& t) {
    return IpaAsProof::read(vk, instances, t);
  }

  // accumulation.rs:45-79: C = sum alpha^i [U_i] (+ omega [S]),  v = sum alpha^i h_i(z)
  static Result<IpaAccumulator> verify(const IpaSuccinctVerifyingKey& vk, const std::vector<IpaAccumulator>& instances,
                                       const IpaAsProof& proof) {
    std::vector<const G1Affine*> u;
    std::vector<Fr> h;
    for (auto& acc : instances) {
      u.push_back(&acc.u);
      h.push_back(h_eval(acc.xi, proof.z));
    }
    if (proof.a_b_u) {
      u.push_back(&proof.a_b_u->u);
      h.push_back(proof.a_b_u->a * proof.z + proof.a_b_u->b);
    }
    const std::vector<Fr> powers_of_alpha = proof.alpha.powers(u.size());
    std::vector<MsmT> terms;
    for (size_t i = 0; i < u.size(); ++i) terms.push_back(MsmT::base(u[i]) * powers_of_alpha[i]);
    MsmT c = MsmT::sum(terms);
    if (proof.omega) {
      if (!vk.s) throw Panic("IpaAs proof carries omega but the key has no S (reference: unwrap, accumulation.rs:73)");
      c += MsmT::base(&*vk.s) * *proof.omega;
    }
    std::vector<std::pair<Fr, Fr>> prods;
    for (size_t i = 0; i < h.size(); ++i) prods.emplace_back(powers_of_alpha[i], h[i]);
    Fr v = L::sum_products(prods);
    return Ipa::succinct_verify(vk, c, proof.z, v, proof.ipa);
  }

  // decider.rs:47-55: U == commit(G, h).  h_coeffs (ipa.rs:405-421) is built on the device straight
  // into the scalar buffer of ONE 2^k-term Pippenger over the resident committing key.
  static Error decide(const IpaDecidingKey& dk, const IpaAccumulator& acc) { return decide_all(dk, {acc}); }

  // decider.rs:57-66: every accumulator must pass (same verdict as the reference's early exit)
  static Error decide_all(const IpaDecidingKey& dk, const std::vector<IpaAccumulator>& accs) {
    if (accs.empty()) return Error{};
    const size_t k = dk.svk.k;
    std::vector<uint8_t> xi(accs.size() * k * 32), u(accs.size() * 64), ok(accs.size());
    for (size_t a = 0; a < accs.size(); ++a) {
      if (accs[a].xi.empty()) throw Panic("h_coeffs of no challenges (reference: assert!, ipa.rs:406)");
      if (accs[a].xi.size() != k)
        throw Panic("IpaAccumulator with xi.len() != k (reference: assert_eq!(scalars.len(), bases.len()), msm.rs:309)");
      for (size_t j = 0; j < k; ++j) accs[a].xi[j].to_bytes(&xi[(a * k + j) * 32]);
      memcpy(&u[64 * a], accs[a].u.b, 64);
    }
    snarkv_ipa_dk* h = dk.handle();
    DeviceScope lock;
    int rc = SNARKV_DEV(ipa_decide_batch)(h, xi.data(), u.data(), accs.size(), ok.data());
    if (rc != SNARKV_OK) throw std::runtime_error(std::string("ipa_decide_batch: ") + SNARKV_DEV_LAST_ERROR());
    for (uint8_t b : ok)
      if (!b) return Error::assertion("U == commit(G, h)");
    return Error{};
  }
};

// ------------------------------------------------------------------ Bgh19
// bgh19.rs:98-153.  The IPA part arrives in halo2's order (S, xi, z, rounds, c, blind, G) and
// is stored as the `IpaProof` it maps to (bgh19.rs:150).
struct Bgh19 {};
struct Bgh19Proof {
  Fr x_1, x_2;
  G1Affine f;
  Fr x_3;
  std::vector<Fr> q_evals;
  Fr x_4;
  IpaProof ipa;

  static Result<Bgh19Proof> read(const IpaSuccinctVerifyingKey& svk, const std::vector<Query<Fr>>& queries, Transcript& t) {
    using Res = Result<Bgh19Proof>;
    Bgh19Proof p;
    p.x_1 = t.squeeze_challenge();
    p.x_2 = t.squeeze_challenge();
    auto f = t.read_ec_point();
    if (!f.ok()) return Res::Err(f.err);
    p.f = *f.value;
    p.x_3 = t.squeeze_challenge();
    const size_t n_sets = bdfg21::query_sets(queries).size();  // the same grouping as bdfg21.rs:121-171 (bgh19.rs:155-215)
    for (size_t i = 0; i < n_sets; ++i) {
      auto q = t.read_scalar();
      if (!q.ok()) return Res::Err(q.err);
      p.q_evals.push_back(*q.value);
    }
    p.x_4 = t.squeeze_challenge();
    auto s = t.read_ec_point();
    if (!s.ok()) return Res::Err(s.err);
    Fr xi = t.squeeze_challenge();
    Fr z = t.squeeze_challenge();
    p.ipa.c_bar_alpha = std::make_pair(*s.value, xi);
    p.ipa.xi_0 = z;
    for (size_t i = 0; i < svk.k; ++i) {
      auto l = t.read_ec_point();
      if (!l.ok()) return Res::Err(l.err);
      auto r = t.read_ec_point();
      if (!r.ok()) return Res::Err(r.err);
      p.ipa.rounds.push_back(IpaProof::Round{*l.value, *r.value, t.squeeze_challenge()});
    }
    auto c = t.read_scalar();
    if (!c.ok()) return Res::Err(c.err);
    auto blind = t.read_scalar();
    if (!blind.ok()) return Res::Err(blind.err);
    auto g = t.read_ec_point();
    if (!g.ok()) return Res::Err(g.err);
    p.ipa.c = *c.value;
    p.ipa.omega_prime = *blind.value;
    p.ipa.u = *g.value;
    return Res::Ok(std::move(p));
  }
};

namespace bgh19 {
// bgh19.rs:313-399 after its two batch inversions (values, not Fractions)
struct QuerySetCoeff {
  std::vector<Fr> eval_coeffs;  // barycentric weights
  Fr r_eval_coeff;              // 1 / sum of the weights
  Fr f_eval_coeff;              // 1 / prod (x_3 - x shift)
};

// bgh19.rs:217-250
inline std::vector<QuerySetCoeff> query_set_coeffs(const std::vector<bdfg21::QuerySet>& sets, const Fr& x, const Fr& x_3) {
  size_t size = 2;
  for (auto& s : sets) size = std::max(size, s.shifts.size());
  const auto powers_of_x = x.powers(size);
  std::vector<QuerySetCoeff> coeffs;
  for (auto& set : sets) {
    QuerySetCoeff c;
    const auto& shifts = set.shifts;
    const Fr& xk1 = powers_of_x[shifts.size() - 1];
    for (size_t j = 0; j < shifts.size(); ++j) {
      Fr ell = Fr::one();  // normalized_ell_prime, bgh19.rs:327-340
      for (size_t i = 0; i < shifts.size(); ++i)
        if (i != j) ell *= (shifts[j] - shifts[i]);
      c.eval_coeffs.push_back(ell * xk1 * x_3 - ell * shifts[j] * xk1 * powers_of_x[1]);  // bgh19.rs:345-355
    }
    c.f_eval_coeff = Fr::one();
    for (auto& sh : shifts) c.f_eval_coeff *= (x_3 - x * sh);  // bgh19.rs:357-364
    coeffs.push_back(std::move(c));
  }
  {
    std::vector<Fr*> denoms;  // first batch inversion (bgh19.rs:245)
    for (auto& c : coeffs) {
      for (auto& e : c.eval_coeffs) denoms.push_back(&e);
      denoms.push_back(&c.f_eval_coeff);
    }
    L::batch_invert(denoms);
  }
  {
    std::vector<Fr*> denoms;  // second: the barycentric-weight sums (bgh19.rs:246, :384-394)
    for (auto& c : coeffs) {
      Fr sum;
      for (auto& e : c.eval_coeffs) sum += e;
      c.r_eval_coeff = sum;
      denoms.push_back(&c.r_eval_coeff);
    }
    L::batch_invert(denoms);
  }
  return coeffs;
}

// the `p` of bgh19.rs:61-93: the commitment whose opening at x_3 must be 0
inline MsmT final_msm(const G1Affine* g0, const std::vector<MsmT>& commitments, const Fr& x,
                      const std::vector<Query<Fr>>& queries, const Bgh19Proof& proof) {
  const auto sets = bdfg21::query_sets(queries);
  if (sets.size() != proof.q_evals.size()) throw Panic("Bgh19: queries changed between read_proof and verify");
  const auto coeffs = bgh19::query_set_coeffs(sets, x, proof.x_3);
  size_t maxp = 0;
  for (auto& s : sets) maxp = std::max(maxp, s.polys.size());
  const auto powers_of_x_1 = proof.x_1.powers(maxp);
  const auto powers_of_x_2 = proof.x_2.powers(sets.size());
  std::vector<Fr> f_evals;  // QuerySet::f_eval, bgh19.rs:276-300
  for (size_t k = 0; k < sets.size(); ++k) {
    const auto& set = sets[k];
    const auto& co = coeffs[k];
    Fr r_eval;
    const size_t np = set.polys.size();
    for (size_t i = 0; i < np; ++i) {
      Fr r_i;
      for (size_t j = 0; j < co.eval_coeffs.size(); ++j) r_i += co.eval_coeffs[j] * *set.evals[i][j];
      r_eval += r_i * co.r_eval_coeff * powers_of_x_1[np - 1 - i];  // r_evals.rev() zip powers_of_x_1
    }
    f_evals.push_back((proof.q_evals[k] - r_eval) * co.f_eval_coeff);
  }
  Fr f_eval;
  for (size_t j = 0; j < sets.size(); ++j) f_eval += powers_of_x_2[j] * f_evals[sets.size() - 1 - j];
  const auto powers_of_x_4 = proof.x_4.powers(sets.size() + 1);
  std::vector<MsmT> terms;
  terms.push_back((MsmT::base(&proof.f) - MsmT::from_constant(f_eval)) * powers_of_x_4[sets.size()]);
  for (size_t k = 0; k < sets.size(); ++k) {
    const auto& set = sets[k];
    std::vector<MsmT> polys;  // QuerySet::msm, bgh19.rs:262-274
    for (size_t m = 0; m < set.polys.size(); ++m)
      polys.push_back(commitments[set.polys[set.polys.size() - 1 - m]] * powers_of_x_1[m]);
    terms.push_back((MsmT::sum(polys) - MsmT::from_constant(proof.q_evals[k])) * powers_of_x_4[sets.size() - 1 - k]);
  }
  auto [msm, constant] = MsmT::sum(terms).split();
  if (constant) msm += MsmT::base(g0) * *constant;
  return msm;
}
}  // namespace bgh19

// `impl PolynomialCommitmentScheme for IpaAs<C, Bgh19>` (bgh19.rs:26-96)
struct IpaBgh19 {
  using VerifyingKey = IpaSuccinctVerifyingKey;
  using Proof = Bgh19Proof;
  using Output = IpaAccumulator;
  static Result<Bgh19Proof> read_proof(const IpaSuccinctVerifyingKey& svk, const std::vector<Query<Fr>>& queries,
                                       Transcript& t) {
    return Bgh19Proof::read(svk, queries, t);
  }
  static Result<IpaAccumulator> verify(const IpaSuccinctVerifyingKey& svk, const std::vector<MsmT>& commitments,
                                       const Fr& x, const std::vector<Query<Fr>>& queries, const Bgh19Proof& proof) {
    const G1Affine g = L::ec_point_load_const(svk.g);
    MsmT p = bgh19::final_msm(&g, commitments, x, queries, proof);
    return Ipa::succinct_verify(svk, p, proof.x_3, L::load_zero(), proof.ipa);
  }
  // the host half only (for batches: Ipa::finish_batch launches all of them together)
  static Ipa::Pending verify_pairs(const IpaSuccinctVerifyingKey& svk, const std::vector<MsmT>& commitments, const Fr& x,
                                   const std::vector<Query<Fr>>& queries, const Bgh19Proof& proof) {
    const G1Affine g = L::ec_point_load_const(svk.g);
    MsmT p = bgh19::final_msm(&g, commitments, x, queries, proof);
    return Ipa::succinct_verify_pairs(svk, p, proof.x_3, L::load_zero(), proof.ipa);
  }
};

}  // namespace snarkv_host
