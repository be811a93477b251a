// Packed wire format of a `PlonkProtocol`, of instance sets and of proof batches: what crosses the C API of
// the host mirror (include/snarkv_host.h).  The reference serialises `PlonkProtocol` through serde (bincode /
// JSON of external-crate types, snark-verifier/src/verifier/plonk/protocol.rs:19-71 under `derive_serde`);
// that form is read by host/serde_json.hpp.  This one is the compact little-endian form documented in
// include/snarkv_host.h and written by tests/plonk_synth.py::pack_protocol:
//
//   protocol := u32 k | Fr gen (domain: 2^k rows, generator) | vec<G1> preprocessed | vec<u32> num_instance |
//               vec<u32> num_witness | vec<u32> num_challenge | vec<(u32 poly, i32 rot)> evaluations |
//               vec<(u32, i32)> queries | u32 chunk_degree | u32 num_chunk | expr numerator |
//               u8 has_initial_state [Fr] | u8 has_ick [vec<G1> bases, u8 has_const [G1]] | u8 linearization |
//               vec<vec<(u32, u32)>> accumulator_indices
//   expr     := u8 tag, then by tag: 0 Constant Fr | 1 Identity | 2 Lagrange i32 | 3 Polynomial u32 i32 |
//               4 Challenge u32 | 5 Negated expr | 6 Sum expr expr | 7 Product expr expr | 8 Scaled expr Fr |
//               9 DistributePowers u32 n, n x expr, expr base
//   instances:= u32 columns, per column u32 m, m x Fr          (one block per proof, blocks concatenated)
//   proofs   := per proof u32 len, len bytes                   (concatenated)
//   (vec<T> = u32 count, count x T; Fr = 32 B LE canonical; G1 = 64 B x|y LE canonical)
#pragma once
#include <cstring>
#include <memory>
#include <vector>

#include "plonk.hpp"

namespace snarkv_host {
namespace wire {

struct PReader {
  const uint8_t* p;
  const uint8_t* end;
  void need(size_t n) {
    if ((size_t)(end - p) < n) throw Panic("truncated protocol bytes");
  }
  uint8_t u8() {
    need(1);
    return *p++;
  }
  uint32_t u32() {
    need(4);
    uint32_t v;
    memcpy(&v, p, 4);
    p += 4;
    return v;
  }
  int32_t i32() {
    need(4);
    int32_t v;
    memcpy(&v, p, 4);
    p += 4;
    return v;
  }
  Fr fr() {
    need(32);
    Fr x;
    if (!Fr::from_bytes(p, &x)) throw Panic("non-canonical Fr in protocol bytes");
    p += 32;
    return x;
  }
  G1Affine g1() {
    need(64);
    G1Affine x = G1Affine::from_bytes(p);
    p += 64;
    return x;
  }
};

// depth-bounded like the serde-JSON / bincode readers (serde_json.hpp): a buffer of repeated tag 5 (Negated) would
// otherwise recurse once per byte and overflow the stack instead of returning SNARKV_HOST_ERR_PANIC
constexpr int kMaxExprDepth = 2048;
inline ExprPtr parse_expr(PReader& rd, int depth = 0) {
  if (depth > kMaxExprDepth) throw Panic("expression nested deeper than 2048 levels");
  auto e = std::make_shared<Expression>();
  switch (rd.u8()) {
    case 0: e->kind = Expression::Constant; e->scalar = rd.fr(); break;
    case 1: e->kind = Expression::Identity; break;
    case 2: e->kind = Expression::Lagrange; e->lagrange = rd.i32(); break;
    case 3: e->kind = Expression::Polynomial; e->query.poly = rd.u32(); e->query.rotation = rd.i32(); break;
    case 4: e->kind = Expression::Challenge; e->index = rd.u32(); break;
    case 5: e->kind = Expression::Negated; e->ch.push_back(parse_expr(rd, depth + 1)); break;
    case 6: e->kind = Expression::Sum; e->ch.push_back(parse_expr(rd, depth + 1)); e->ch.push_back(parse_expr(rd, depth + 1)); break;
    case 7: e->kind = Expression::Product; e->ch.push_back(parse_expr(rd, depth + 1)); e->ch.push_back(parse_expr(rd, depth + 1)); break;
    case 8: e->kind = Expression::Scaled; e->ch.push_back(parse_expr(rd, depth + 1)); e->scalar = rd.fr(); break;
    case 9: {
      e->kind = Expression::DistributePowers;
      uint32_t n = rd.u32();
      for (uint32_t i = 0; i < n; ++i) e->ch.push_back(parse_expr(rd, depth + 1));
      e->ch.push_back(parse_expr(rd, depth + 1));
      break;
    }
    default: throw Panic("bad expression tag");
  }
  return e;
}

inline PlonkProtocol parse_protocol(const uint8_t* b, size_t len) {
  PReader rd{b, b + len};
  PlonkProtocol pr;
  uint32_t k = rd.u32();
  pr.domain = Domain::make(k, rd.fr());
  for (uint32_t n = rd.u32(), i = 0; i < n; ++i) pr.preprocessed.push_back(rd.g1());
  for (auto* v : {&pr.num_instance, &pr.num_witness, &pr.num_challenge})
    for (uint32_t n = rd.u32(), i = 0; i < n; ++i) v->push_back(rd.u32());
  for (auto* v : {&pr.evaluations, &pr.queries})
    for (uint32_t n = rd.u32(), i = 0; i < n; ++i) {
      PQuery q;
      q.poly = rd.u32();
      q.rotation = rd.i32();
      v->push_back(q);
    }
  pr.quotient.chunk_degree = rd.u32();
  pr.quotient.num_chunk = rd.u32();
  pr.quotient.numerator = parse_expr(rd);
  if (rd.u8()) pr.transcript_initial_state = rd.fr();
  if (rd.u8()) {
    InstanceCommittingKey ick;
    for (uint32_t n = rd.u32(), i = 0; i < n; ++i) ick.bases.push_back(rd.g1());
    if (rd.u8()) ick.constant = rd.g1();
    pr.instance_committing_key = ick;
  }
  uint8_t lin = rd.u8();
  if (lin > 2) throw Panic("bad linearization tag");
  pr.linearization = lin == 0 ? Linearization::None : lin == 1 ? Linearization::WithoutConstant : Linearization::MinusVanishingTimesQuotient;
  for (uint32_t n = rd.u32(), i = 0; i < n; ++i) {
    std::vector<std::pair<size_t, size_t>> idx;
    for (uint32_t m = rd.u32(), j = 0; j < m; ++j) {
      uint32_t a = rd.u32(), c = rd.u32();
      idx.emplace_back(a, c);
    }
    pr.accumulator_indices.push_back(idx);
  }
  if (rd.p != rd.end) throw Panic("trailing protocol bytes");
  return pr;
}

inline std::vector<std::vector<Fr>> parse_instances(const uint8_t* b, size_t len) {
  PReader rd{b, b + len};
  std::vector<std::vector<Fr>> out;
  for (uint32_t n = rd.u32(), i = 0; i < n; ++i) {
    std::vector<Fr> v;
    for (uint32_t m = rd.u32(), j = 0; j < m; ++j) v.push_back(rd.fr());
    out.push_back(v);
  }
  return out;
}

// N instance blocks + N length-prefixed proofs -> per-proof vectors (what `Aggregator` / `verify_batch` take)
inline void split_batch(const uint8_t* instances, size_t ilen, const uint8_t* proofs, size_t prlen, uint32_t n,
                        std::vector<std::vector<std::vector<Fr>>>& insts, std::vector<std::vector<uint8_t>>& pbytes) {
  const uint8_t* ip = instances;
  const uint8_t* pp = proofs;
  for (uint32_t i = 0; i < n; ++i) {
    PReader rd{ip, instances + ilen};  // each packed block is self-delimiting
    uint32_t cols = rd.u32();
    for (uint32_t c = 0; c < cols; ++c) {
      uint32_t m = rd.u32();
      rd.need(32 * (size_t)m);
      rd.p += 32 * (size_t)m;
    }
    insts.push_back(parse_instances(ip, (size_t)(rd.p - ip)));
    ip = rd.p;
    if ((size_t)(proofs + prlen - pp) < 4) throw Panic("truncated proofs");
    uint32_t len;
    memcpy(&len, pp, 4);
    if ((size_t)(proofs + prlen - pp) < 4 + (size_t)len) throw Panic("proof length runs past the buffer");
    pbytes.emplace_back(pp + 4, pp + 4 + len);
    pp += 4 + len;
  }
}

// ---- writers (the inverse of the parsers above)
struct PWriter {
  std::vector<uint8_t> b;
  void u8(uint8_t v) { b.push_back(v); }
  void u32(uint32_t v) { b.insert(b.end(), (uint8_t*)&v, (uint8_t*)&v + 4); }
  void i32(int32_t v) { b.insert(b.end(), (uint8_t*)&v, (uint8_t*)&v + 4); }
  void fr(const Fr& x) {
    uint8_t t[32];
    x.to_bytes(t);
    b.insert(b.end(), t, t + 32);
  }
  void g1(const G1Affine& p) { b.insert(b.end(), p.b, p.b + 64); }
};

inline void pack_expr(PWriter& w, const Expression& e) {
  switch (e.kind) {
    case Expression::Constant: w.u8(0); w.fr(e.scalar); break;
    case Expression::Identity: w.u8(1); break;
    case Expression::Lagrange: w.u8(2); w.i32(e.lagrange); break;
    case Expression::Polynomial: w.u8(3); w.u32((uint32_t)e.query.poly); w.i32(e.query.rotation); break;
    case Expression::Challenge: w.u8(4); w.u32((uint32_t)e.index); break;
    case Expression::Negated: w.u8(5); pack_expr(w, *e.ch.at(0)); break;
    case Expression::Sum: w.u8(6); pack_expr(w, *e.ch.at(0)); pack_expr(w, *e.ch.at(1)); break;
    case Expression::Product: w.u8(7); pack_expr(w, *e.ch.at(0)); pack_expr(w, *e.ch.at(1)); break;
    case Expression::Scaled: w.u8(8); pack_expr(w, *e.ch.at(0)); w.fr(e.scalar); break;
    case Expression::DistributePowers:
      w.u8(9);
      if (e.ch.empty()) throw Panic("DistributePowers without a base expression");
      w.u32((uint32_t)(e.ch.size() - 1));
      for (auto& c : e.ch) pack_expr(w, *c);
      break;
  }
}

inline std::vector<uint8_t> pack_protocol(const PlonkProtocol& pr) {
  PWriter w;
  w.u32((uint32_t)pr.domain.k);
  w.fr(pr.domain.gen);
  w.u32((uint32_t)pr.preprocessed.size());
  for (auto& p : pr.preprocessed) w.g1(p);
  for (auto* v : {&pr.num_instance, &pr.num_witness, &pr.num_challenge}) {
    w.u32((uint32_t)v->size());
    for (size_t x : *v) w.u32((uint32_t)x);
  }
  for (auto* v : {&pr.evaluations, &pr.queries}) {
    w.u32((uint32_t)v->size());
    for (auto& q : *v) {
      w.u32((uint32_t)q.poly);
      w.i32(q.rotation);
    }
  }
  w.u32((uint32_t)pr.quotient.chunk_degree);
  w.u32((uint32_t)pr.quotient.num_chunk);
  pack_expr(w, *pr.quotient.numerator);
  w.u8(pr.transcript_initial_state ? 1 : 0);
  if (pr.transcript_initial_state) w.fr(*pr.transcript_initial_state);
  w.u8(pr.instance_committing_key ? 1 : 0);
  if (pr.instance_committing_key) {
    w.u32((uint32_t)pr.instance_committing_key->bases.size());
    for (auto& p : pr.instance_committing_key->bases) w.g1(p);
    w.u8(pr.instance_committing_key->constant ? 1 : 0);
    if (pr.instance_committing_key->constant) w.g1(*pr.instance_committing_key->constant);
  }
  w.u8(pr.linearization == Linearization::None ? 0 : pr.linearization == Linearization::WithoutConstant ? 1 : 2);
  w.u32((uint32_t)pr.accumulator_indices.size());
  for (auto& idx : pr.accumulator_indices) {
    w.u32((uint32_t)idx.size());
    for (auto& t : idx) {
      w.u32((uint32_t)t.first);
      w.u32((uint32_t)t.second);
    }
  }
  return std::move(w.b);
}

inline std::vector<uint8_t> pack_instances(const std::vector<std::vector<Fr>>& inst) {
  PWriter w;
  w.u32((uint32_t)inst.size());
  for (auto& col : inst) {
    w.u32((uint32_t)col.size());
    for (auto& x : col) w.fr(x);
  }
  return std::move(w.b);
}

}  // namespace wire
}  // namespace snarkv_host
