// Interchange with the reference's own serialisations:
//
//   serde_json / bincode of `PlonkProtocol<G1Affine>`    snark-verifier/src/verifier/plonk/protocol.rs:17-72 (derive_serde)
//     with `Domain` (util/arithmetic.rs:120-134), `Rotation(i32)` (:92-95), `Query` (protocol.rs:302-307),
//     `QuotientPolynomial` (:286-294), `Expression` (:318-330), `CommonPolynomial` (:191-196),
//     `LinearizationStrategy` (:529-540), `InstanceCommittingKey` (:542-547)
//   bincode of the SDK's `Snark { protocol, instances, proof }`   snark-verifier-sdk/src/lib.rs:47-53,
//     written by `gen_snark` (snark-verifier-sdk/src/halo2.rs:266-281) and read by `read_snark` (:313-316)
//
// What serde's derive fixes (structs = field order / field names, enums = externally tagged in JSON and a u32
// variant index in bincode, Option = null / u8 tag, Vec = u64 length prefix in bincode, usize = u64) is
// implemented as such.  What it does NOT fix is how the external crate halo2curves 0.6.0 serialises `Fr`, `Fq` and
// `G1Affine` under its `derive_serde` feature (not vendored; no Rust here): either the crate's 32-byte canonical
// little-endian form (hex string in human-readable formats) or a plain derive on the limb array (4 x u64,
// Montgomery form).  Both are accepted: strings are canonical LE hex; 4-element arrays / raw 32-byte blobs are
// disambiguated by the one self-checking value every protocol carries: `domain.n_inv * n == 1`.
// UNPINNED until tools/refgen (which needs a Rust toolchain) has produced a real dump; the loaders are exercised
// against samples written from the struct definitions (tests/test_interchange.py).
#pragma once
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "plonk.hpp"

namespace snarkv_host {
namespace interchange {

// ---- field encodings -------------------------------------------------------------------------------------
// Fq (BN254 base field) is only ever carried as bytes on the host; the one operation needed here is
// Montgomery limbs -> canonical (REDC by R = 2^256) for the "derive on the limb array" hypothesis.
inline void fq_from_montgomery(const uint64_t in[4], uint8_t out[32]) {
  static const uint64_t P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  const uint64_t PINV = 0x87d20782e4866389ull;
  uint64_t t[5] = {in[0], in[1], in[2], in[3], 0};
  for (int i = 0; i < 4; ++i) {
    uint64_t m = t[0] * PINV;
    unsigned __int128 c = ((unsigned __int128)m * P[0] + t[0]) >> 64;
    for (int j = 1; j < 4; ++j) {
      c += (unsigned __int128)m * P[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = (uint64_t)(c >> 64);
  }
  bool ge = true;  // t >= p ?
  for (int i = 3; i >= 0; --i)
    if (t[i] != P[i]) {
      ge = t[i] > P[i];
      break;
    }
  if (ge) {
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 x = (unsigned __int128)t[i] - P[i] - (uint64_t)br;
      t[i] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
  }
  memcpy(out, t, 32);
}

struct FieldCodec {
  bool montgomery = false;  // raw 32-byte blobs / 4-limb arrays hold Montgomery residues (R = 2^256)
  Fr fr(const uint8_t b[32]) const {
    Fr x;
    if (montgomery) {
      memcpy(x.v, b, 32);
      uint8_t chk[32];
      x.to_bytes(chk);  // any 256-bit pattern < r is a valid residue; reject the rest like `from_repr`
      Fr back;
      if (!Fr::from_bytes(chk, &back) || !(back == x)) throw Panic("interchange: Fr limbs out of range");
      return x;
    }
    if (!Fr::from_bytes(b, &x)) throw Panic("interchange: non-canonical Fr");
    return x;
  }
  G1Affine g1(const uint8_t b[64]) const {
    G1Affine p;
    if (montgomery) {
      uint64_t l[4];
      memcpy(l, b, 32);
      fq_from_montgomery(l, p.b);
      memcpy(l, b + 32, 32);
      fq_from_montgomery(l, p.b + 32);
    } else {
      memcpy(p.b, b, 64);
    }
    return p;
  }
  // decide the encoding from a domain: exactly one reading satisfies n_inv * n == 1
  static FieldCodec detect(uint64_t n, const uint8_t n_inv[32]) {
    for (bool mont : {false, true}) {
      FieldCodec c;
      c.montgomery = mont;
      try {
        if (c.fr(n_inv) * Fr::from_u64(n) == Fr::one()) return c;
      } catch (const Panic&) {
      }
    }
    throw Panic("interchange: domain.n_inv is not the inverse of n under either field encoding");
  }
};

// ---- a small JSON reader (numbers kept as text: u64 limbs must not go through double) --------------------
struct JValue {
  enum Type { Null, Bool, Num, Str, Arr, Obj } t = Null;
  bool b = false;
  std::string s;  // Num: the literal; Str: the unescaped string
  std::vector<JValue> a;
  std::vector<std::pair<std::string, JValue>> o;
  const JValue& at(const char* key) const {
    if (t != Obj) throw Panic(std::string("json: expected an object holding '") + key + "'");
    for (auto& kv : o)
      if (kv.first == key) return kv.second;
    throw Panic(std::string("json: missing field '") + key + "'");
  }
  const JValue* find(const char* key) const {
    if (t != Obj) return nullptr;
    for (auto& kv : o)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  uint64_t u64() const {
    if (t != Num || s.empty() || s[0] == '-') throw Panic("json: expected an unsigned integer");
    char* end = nullptr;
    errno = 0;
    unsigned long long v = strtoull(s.c_str(), &end, 10);
    if (errno || *end) throw Panic("json: bad unsigned integer '" + s + "'");
    return (uint64_t)v;
  }
  int64_t i64() const {
    if (t != Num) throw Panic("json: expected an integer");
    char* end = nullptr;
    errno = 0;
    long long v = strtoll(s.c_str(), &end, 10);
    if (errno || *end) throw Panic("json: bad integer '" + s + "'");
    return (int64_t)v;
  }
  const std::vector<JValue>& arr() const {
    if (t != Arr) throw Panic("json: expected an array");
    return a;
  }
};

class JsonReader {
 public:
  JsonReader(const uint8_t* p, size_t n) : p_((const char*)p), end_((const char*)p + n) {}
  JValue parse_document() {
    JValue v = value(0);
    ws();
    if (p_ != end_) throw Panic("json: trailing characters");
    return v;
  }

 private:
  const char *p_, *end_;
  void ws() {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
  }
  char peek() {
    ws();
    if (p_ >= end_) throw Panic("json: unexpected end");
    return *p_;
  }
  void expect(char c) {
    if (peek() != c) throw Panic(std::string("json: expected '") + c + "'");
    ++p_;
  }
  bool lit(const char* w) {
    size_t n = strlen(w);
    if ((size_t)(end_ - p_) >= n && !memcmp(p_, w, n)) {
      p_ += n;
      return true;
    }
    return false;
  }
  std::string string() {
    expect('"');
    std::string out;
    while (true) {
      if (p_ >= end_) throw Panic("json: unterminated string");
      char c = *p_++;
      if (c == '"') break;
      if (c == '\\') {
        if (p_ >= end_) throw Panic("json: bad escape");
        char e = *p_++;
        switch (e) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            if (end_ - p_ < 4) throw Panic("json: bad \\u escape");
            unsigned cp = (unsigned)strtoul(std::string(p_, p_ + 4).c_str(), nullptr, 16);
            p_ += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += e;  // \" \\ \/
        }
      } else {
        out += c;
      }
    }
    return out;
  }
  JValue value(int depth) {
    if (depth > 4096) throw Panic("json: nesting too deep");
    JValue v;
    char c = peek();
    if (c == '{') {
      ++p_;
      v.t = JValue::Obj;
      if (peek() == '}') { ++p_; return v; }
      while (true) {
        ws();
        std::string k = string();
        expect(':');
        v.o.emplace_back(std::move(k), value(depth + 1));
        if (peek() == ',') { ++p_; continue; }
        expect('}');
        break;
      }
    } else if (c == '[') {
      ++p_;
      v.t = JValue::Arr;
      if (peek() == ']') { ++p_; return v; }
      while (true) {
        v.a.push_back(value(depth + 1));
        if (peek() == ',') { ++p_; continue; }
        expect(']');
        break;
      }
    } else if (c == '"') {
      v.t = JValue::Str;
      v.s = string();
    } else if (lit("null")) {
      v.t = JValue::Null;
    } else if (lit("true")) {
      v.t = JValue::Bool;
      v.b = true;
    } else if (lit("false")) {
      v.t = JValue::Bool;
    } else {
      v.t = JValue::Num;
      const char* s = p_;
      while (p_ < end_ && (*p_ == '-' || *p_ == '+' || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || (*p_ >= '0' && *p_ <= '9'))) ++p_;
      if (s == p_) throw Panic("json: unexpected character");
      v.s.assign(s, p_);
    }
    return v;
  }
};

// a field element / point as 32 / 64 raw bytes in the SOURCE encoding (decoded later through the FieldCodec)
inline void json_field_raw(const JValue& v, uint8_t out[32]) {
  if (v.t == JValue::Str) {  // canonical LE hex ("0x" tolerated)
    std::string h = v.s;
    if (h.size() >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) h = h.substr(2);
    if (h.size() != 64) throw Panic("json: field element hex must be 32 bytes");
    for (int i = 0; i < 32; ++i) {
      auto nib = [](char c) -> int {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        throw Panic("json: bad hex digit");
      };
      out[i] = (uint8_t)(nib(h[2 * i]) << 4 | nib(h[2 * i + 1]));
    }
    return;
  }
  const auto& a = v.arr();
  if (a.size() == 4) {  // [u64; 4] limbs, little-endian limb order
    for (int i = 0; i < 4; ++i) {
      uint64_t l = a[i].u64();
      memcpy(out + 8 * i, &l, 8);
    }
    return;
  }
  if (a.size() == 32) {  // [u8; 32]
    for (int i = 0; i < 32; ++i) {
      uint64_t b = a[i].u64();
      if (b > 255) throw Panic("json: byte out of range");
      out[i] = (uint8_t)b;
    }
    return;
  }
  throw Panic("json: field element must be a hex string, 4 limbs or 32 bytes");
}
inline bool json_field_is_string(const JValue& v) { return v.t == JValue::Str; }

struct JsonCtx {
  FieldCodec limbs;  // for array-encoded elements; strings are always canonical
  Fr fr(const JValue& v) const {
    uint8_t b[32];
    json_field_raw(v, b);
    return json_field_is_string(v) || (v.t == JValue::Arr && v.a.size() == 32) ? FieldCodec{}.fr(b) : limbs.fr(b);
  }
  G1Affine g1(const JValue& v) const {
    uint8_t b[64];
    const JValue &x = v.at("x"), &y = v.at("y");
    json_field_raw(x, b);
    json_field_raw(y, b + 32);
    return json_field_is_string(x) || (x.t == JValue::Arr && x.a.size() == 32) ? FieldCodec{}.g1(b) : limbs.g1(b);
  }
};

inline PQuery json_query(const JValue& v) {
  PQuery q;
  q.poly = (size_t)v.at("poly").u64();
  q.rotation = (int32_t)v.at("rotation").i64();  // Rotation(pub i32): a newtype struct serialises as its field
  return q;
}

inline ExprPtr json_expr(const JValue& v, const JsonCtx& cx, int depth = 0) {
  if (depth > 2048) throw Panic("json: expression nesting too deep");
  auto e = std::make_shared<Expression>();
  if (v.t != JValue::Obj || v.o.size() != 1) throw Panic("json: an Expression is a single-key object (externally tagged enum)");
  const std::string& tag = v.o[0].first;
  const JValue& body = v.o[0].second;
  auto sub = [&](const JValue& x) { return json_expr(x, cx, depth + 1); };
  if (tag == "Constant") {
    e->kind = Expression::Constant;
    e->scalar = cx.fr(body);
  } else if (tag == "CommonPolynomial") {
    if (body.t == JValue::Str && body.s == "Identity") {
      e->kind = Expression::Identity;
    } else if (body.t == JValue::Obj && body.find("Lagrange")) {
      e->kind = Expression::Lagrange;
      e->lagrange = (int32_t)body.at("Lagrange").i64();
    } else {
      throw Panic("json: unknown CommonPolynomial");
    }
  } else if (tag == "Polynomial") {
    e->kind = Expression::Polynomial;
    e->query = json_query(body);
  } else if (tag == "Challenge") {
    e->kind = Expression::Challenge;
    e->index = (size_t)body.u64();
  } else if (tag == "Negated") {
    e->kind = Expression::Negated;
    e->ch.push_back(sub(body));
  } else if (tag == "Sum" || tag == "Product") {
    e->kind = tag == "Sum" ? Expression::Sum : Expression::Product;
    if (body.arr().size() != 2) throw Panic("json: Sum / Product take two operands");
    e->ch.push_back(sub(body.a[0]));
    e->ch.push_back(sub(body.a[1]));
  } else if (tag == "Scaled") {
    e->kind = Expression::Scaled;
    if (body.arr().size() != 2) throw Panic("json: Scaled takes (expression, scalar)");
    e->ch.push_back(sub(body.a[0]));
    e->scalar = cx.fr(body.a[1]);
  } else if (tag == "DistributePowers") {
    e->kind = Expression::DistributePowers;
    if (body.arr().size() != 2) throw Panic("json: DistributePowers takes (expressions, base)");
    for (auto& x : body.a[0].arr()) e->ch.push_back(sub(x));
    e->ch.push_back(sub(body.a[1]));
  } else {
    throw Panic("json: unknown Expression variant '" + tag + "'");
  }
  return e;
}

inline PlonkProtocol protocol_from_json(const JValue& v) {
  PlonkProtocol pr;
  const JValue& d = v.at("domain");
  JsonCtx cx;
  {
    const JValue& ninv = d.at("n_inv");
    if (!json_field_is_string(ninv) && ninv.arr().size() == 4) {
      uint8_t b[32];
      json_field_raw(ninv, b);
      cx.limbs = FieldCodec::detect(d.at("n").u64(), b);
    }
  }
  pr.domain.k = (size_t)d.at("k").u64();
  pr.domain.n = (size_t)d.at("n").u64();
  if (pr.domain.k >= 64 || pr.domain.n != ((size_t)1 << pr.domain.k)) throw Panic("json: domain.n != 2^k");
  pr.domain.n_inv = cx.fr(d.at("n_inv"));
  pr.domain.gen = cx.fr(d.at("gen"));
  pr.domain.gen_inv = cx.fr(d.at("gen_inv"));
  if (!(pr.domain.n_inv * Fr::from_u64((uint64_t)pr.domain.n) == Fr::one()) || !(pr.domain.gen * pr.domain.gen_inv == Fr::one()))
    throw Panic("json: inconsistent domain (n_inv / gen_inv)");
  for (auto& p : v.at("preprocessed").arr()) pr.preprocessed.push_back(cx.g1(p));
  for (auto& x : v.at("num_instance").arr()) pr.num_instance.push_back((size_t)x.u64());
  for (auto& x : v.at("num_witness").arr()) pr.num_witness.push_back((size_t)x.u64());
  for (auto& x : v.at("num_challenge").arr()) pr.num_challenge.push_back((size_t)x.u64());
  for (auto& q : v.at("evaluations").arr()) pr.evaluations.push_back(json_query(q));
  for (auto& q : v.at("queries").arr()) pr.queries.push_back(json_query(q));
  const JValue& qt = v.at("quotient");
  pr.quotient.chunk_degree = (size_t)qt.at("chunk_degree").u64();
  pr.quotient.num_chunk = (size_t)qt.at("num_chunk").u64();
  pr.quotient.numerator = json_expr(qt.at("numerator"), cx);
  const JValue& tis = v.at("transcript_initial_state");
  if (tis.t != JValue::Null) pr.transcript_initial_state = cx.fr(tis);
  const JValue& ick = v.at("instance_committing_key");
  if (ick.t != JValue::Null) {
    InstanceCommittingKey k;
    for (auto& p : ick.at("bases").arr()) k.bases.push_back(cx.g1(p));
    const JValue& c = ick.at("constant");
    if (c.t != JValue::Null) k.constant = cx.g1(c);
    pr.instance_committing_key = k;
  }
  const JValue& lin = v.at("linearization");
  if (lin.t == JValue::Null) pr.linearization = Linearization::None;
  else if (lin.t == JValue::Str && lin.s == "WithoutConstant") pr.linearization = Linearization::WithoutConstant;
  else if (lin.t == JValue::Str && lin.s == "MinusVanishingTimesQuotient") pr.linearization = Linearization::MinusVanishingTimesQuotient;
  else throw Panic("json: unknown LinearizationStrategy");
  for (auto& row : v.at("accumulator_indices").arr()) {
    std::vector<std::pair<size_t, size_t>> idx;
    for (auto& t : row.arr()) {
      if (t.arr().size() != 2) throw Panic("json: accumulator index is a pair");
      idx.emplace_back((size_t)t.a[0].u64(), (size_t)t.a[1].u64());
    }
    pr.accumulator_indices.push_back(idx);
  }
  return pr;
}

// the SDK's `Snark` (snark-verifier-sdk/src/lib.rs:47-53)
struct SnarkData {
  PlonkProtocol protocol;
  std::vector<std::vector<Fr>> instances;
  std::vector<uint8_t> proof;
};

inline SnarkData snark_from_json(const JValue& v) {
  SnarkData s;
  s.protocol = protocol_from_json(v.at("protocol"));
  JsonCtx cx;
  {  // instances use the protocol's limb encoding (same crate, same impl)
    const JValue& ninv = v.at("protocol").at("domain").at("n_inv");
    if (!json_field_is_string(ninv) && ninv.arr().size() == 4) {
      uint8_t b[32];
      json_field_raw(ninv, b);
      cx.limbs = FieldCodec::detect(v.at("protocol").at("domain").at("n").u64(), b);
    }
  }
  for (auto& col : v.at("instances").arr()) {
    std::vector<Fr> c;
    for (auto& x : col.arr()) c.push_back(cx.fr(x));
    s.instances.push_back(c);
  }
  for (auto& b : v.at("proof").arr()) {
    uint64_t x = b.u64();
    if (x > 255) throw Panic("json: proof byte out of range");
    s.proof.push_back((uint8_t)x);
  }
  return s;
}

// ---- bincode 1.x, default options: little-endian, fixed-width ints, u64 lengths, u32 enum variants ----------
class BincodeReader {
 public:
  BincodeReader(const uint8_t* p, size_t n) : p_(p), end_(p + n) {}
  FieldCodec codec;
  bool done() const { return p_ == end_; }
  void need(size_t n) const {
    if ((size_t)(end_ - p_) < n) throw Panic("bincode: truncated input");
  }
  uint8_t u8() { need(1); return *p_++; }
  uint32_t u32() { need(4); uint32_t v; memcpy(&v, p_, 4); p_ += 4; return v; }
  int32_t i32() { need(4); int32_t v; memcpy(&v, p_, 4); p_ += 4; return v; }
  uint64_t u64() { need(8); uint64_t v; memcpy(&v, p_, 8); p_ += 8; return v; }
  size_t len(size_t elem_min) {  // a Vec length, sanity-bounded by what is left
    uint64_t n = u64();
    if (elem_min && n > (uint64_t)(end_ - p_) / elem_min) throw Panic("bincode: length runs past the input");
    return (size_t)n;
  }
  bool option() {
    uint8_t t = u8();
    if (t > 1) throw Panic("bincode: bad Option tag");
    return t == 1;
  }
  const uint8_t* raw(size_t n) { need(n); const uint8_t* r = p_; p_ += n; return r; }
  Fr fr() { return codec.fr(raw(32)); }
  G1Affine g1() { return codec.g1(raw(64)); }

 private:
  const uint8_t *p_, *end_;
};

inline PQuery bincode_query(BincodeReader& rd) {
  PQuery q;
  q.poly = (size_t)rd.u64();
  q.rotation = rd.i32();
  return q;
}

inline ExprPtr bincode_expr(BincodeReader& rd, int depth = 0) {
  if (depth > 2048) throw Panic("bincode: expression nesting too deep");
  auto e = std::make_shared<Expression>();
  auto sub = [&] { return bincode_expr(rd, depth + 1); };
  switch (rd.u32()) {  // variant order of protocol.rs:320-330
    case 0: e->kind = Expression::Constant; e->scalar = rd.fr(); break;
    case 1:
      switch (rd.u32()) {  // CommonPolynomial, protocol.rs:193-196
        case 0: e->kind = Expression::Identity; break;
        case 1: e->kind = Expression::Lagrange; e->lagrange = rd.i32(); break;
        default: throw Panic("bincode: unknown CommonPolynomial variant");
      }
      break;
    case 2: e->kind = Expression::Polynomial; e->query = bincode_query(rd); break;
    case 3: e->kind = Expression::Challenge; e->index = (size_t)rd.u64(); break;
    case 4: e->kind = Expression::Negated; e->ch.push_back(sub()); break;
    case 5: e->kind = Expression::Sum; e->ch.push_back(sub()); e->ch.push_back(sub()); break;
    case 6: e->kind = Expression::Product; e->ch.push_back(sub()); e->ch.push_back(sub()); break;
    case 7: e->kind = Expression::Scaled; e->ch.push_back(sub()); e->scalar = rd.fr(); break;
    case 8: {
      e->kind = Expression::DistributePowers;
      size_t n = rd.len(4);
      for (size_t i = 0; i < n; ++i) e->ch.push_back(sub());
      e->ch.push_back(sub());
      break;
    }
    default: throw Panic("bincode: unknown Expression variant");
  }
  return e;
}

inline PlonkProtocol bincode_protocol(BincodeReader& rd) {
  PlonkProtocol pr;
  pr.domain.k = (size_t)rd.u64();
  pr.domain.n = (size_t)rd.u64();
  if (pr.domain.k >= 64 || pr.domain.n != ((size_t)1 << pr.domain.k)) throw Panic("bincode: domain.n != 2^k");
  const uint8_t* ninv = rd.raw(32);
  rd.codec = FieldCodec::detect((uint64_t)pr.domain.n, ninv);
  pr.domain.n_inv = rd.codec.fr(ninv);
  pr.domain.gen = rd.fr();
  pr.domain.gen_inv = rd.fr();
  if (!(pr.domain.gen * pr.domain.gen_inv == Fr::one())) throw Panic("bincode: inconsistent domain (gen_inv)");
  for (size_t n = rd.len(64), i = 0; i < n; ++i) pr.preprocessed.push_back(rd.g1());
  for (auto* v : {&pr.num_instance, &pr.num_witness, &pr.num_challenge})
    for (size_t n = rd.len(8), i = 0; i < n; ++i) v->push_back((size_t)rd.u64());
  for (auto* v : {&pr.evaluations, &pr.queries})
    for (size_t n = rd.len(12), i = 0; i < n; ++i) v->push_back(bincode_query(rd));
  pr.quotient.chunk_degree = (size_t)rd.u64();
  pr.quotient.num_chunk = (size_t)rd.u64();
  pr.quotient.numerator = bincode_expr(rd);
  if (rd.option()) pr.transcript_initial_state = rd.fr();
  if (rd.option()) {
    InstanceCommittingKey k;
    for (size_t n = rd.len(64), i = 0; i < n; ++i) k.bases.push_back(rd.g1());
    if (rd.option()) k.constant = rd.g1();
    pr.instance_committing_key = k;
  }
  if (rd.option()) {
    switch (rd.u32()) {
      case 0: pr.linearization = Linearization::WithoutConstant; break;
      case 1: pr.linearization = Linearization::MinusVanishingTimesQuotient; break;
      default: throw Panic("bincode: unknown LinearizationStrategy variant");
    }
  }
  for (size_t n = rd.len(8), i = 0; i < n; ++i) {
    std::vector<std::pair<size_t, size_t>> idx;
    for (size_t m = rd.len(16), j = 0; j < m; ++j) {
      uint64_t a = rd.u64(), b = rd.u64();
      idx.emplace_back((size_t)a, (size_t)b);
    }
    pr.accumulator_indices.push_back(idx);
  }
  return pr;
}

inline SnarkData bincode_snark(const uint8_t* p, size_t n) {
  BincodeReader rd(p, n);
  SnarkData s;
  s.protocol = bincode_protocol(rd);
  for (size_t c = rd.len(8), i = 0; i < c; ++i) {
    std::vector<Fr> col;
    for (size_t m = rd.len(32), j = 0; j < m; ++j) col.push_back(rd.fr());
    s.instances.push_back(col);
  }
  size_t pl = rd.len(1);
  const uint8_t* pb = rd.raw(pl);
  s.proof.assign(pb, pb + pl);
  if (!rd.done()) throw Panic("bincode: trailing bytes after the Snark");
  return s;
}

}  // namespace interchange

namespace serde_json {
inline PlonkProtocol parse_protocol(const uint8_t* p, size_t n) {
  return interchange::protocol_from_json(interchange::JsonReader(p, n).parse_document());
}
inline interchange::SnarkData parse_snark(const uint8_t* p, size_t n) {
  return interchange::snark_from_json(interchange::JsonReader(p, n).parse_document());
}
}  // namespace serde_json
namespace bincode {
inline PlonkProtocol parse_protocol(const uint8_t* p, size_t n) {
  interchange::BincodeReader rd(p, n);
  PlonkProtocol pr = interchange::bincode_protocol(rd);
  if (!rd.done()) throw Panic("bincode: trailing bytes after the PlonkProtocol");
  return pr;
}
inline interchange::SnarkData parse_snark(const uint8_t* p, size_t n) { return interchange::bincode_snark(p, n); }
}  // namespace bincode

}  // namespace snarkv_host
