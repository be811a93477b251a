// Mirror of `util::msm::Msm` (reference snark-verifier/src/util/msm.rs:20-226):
// the deferred linear combination  constant * G + sum scalar_i * base_i.
#pragma once
#include <algorithm>
#include <cstring>
#include <memory>
#include <optional>
#include <unordered_map>
#include <vector>

#include "loader.hpp"

namespace snarkv_host {

template <class L = GpuNativeLoader>
class Msm {
 public:
  using Scalar = typename L::LoadedScalar;
  using Point = typename L::LoadedEcPoint;

  std::optional<Scalar> constant;
  std::vector<Scalar> scalars;
  std::vector<const Point*> bases;  // borrowed, as `&'a L::LoadedEcPoint`

 private:
  static constexpr size_t kIndexThreshold = 48;
  // first 8 bytes of x -> position; built only past the threshold and never copied (a copy rebuilds it on its first
  // `push` past the threshold): the verifier clones Msms freely -- ~90 short-lived ones per proof -- and a hash table
  // member made every one of them cost a constructor / destructor pair
  using Index = std::unordered_multimap<uint64_t, size_t>;
  std::unique_ptr<Index> index_;
  static uint64_t key_of(const Point& p) {
    uint64_t k;
    memcpy(&k, p.b, 8);
    return k;
  }
  void rebuild_index() {
    if (!index_) index_ = std::make_unique<Index>();
    index_->clear();
    index_->reserve(2 * bases.size() + 64);
    for (size_t i = 0; i < bases.size(); ++i) index_->emplace(key_of(*bases[i]), i);
  }

 public:
  Msm() = default;
  Msm(const Msm& o) : constant(o.constant), scalars(o.scalars), bases(o.bases) {}
  Msm(Msm&&) = default;
  Msm& operator=(const Msm& o) {
    if (this != &o) {
      constant = o.constant;
      scalars = o.scalars;
      bases = o.bases;
      index_.reset();
    }
    return *this;
  }
  Msm& operator=(Msm&&) = default;
  // msm.rs:46-51
  static Msm from_constant(const Scalar& c) {
    Msm m;
    m.constant = c;
    return m;
  }
  // msm.rs:54-61
  static Msm base(const Point* b) {
    Msm m;
    m.scalars.push_back(L::load_one());
    m.bases.push_back(b);
    return m;
  }
  size_t size() const { return bases.size(); }
  // msm.rs:72-74
  std::optional<Scalar> try_into_constant() const {
    if (!bases.empty()) return std::nullopt;
    if (!constant) throw Panic("try_into_constant on empty Msm (reference: unwrap, msm.rs:73)");
    return constant;
  }

  // msm.rs:63-66: (self without its constant, the constant)
  std::pair<Msm, std::optional<Scalar>> split() const {
    Msm m = *this;
    std::optional<Scalar> c = m.constant;
    m.constant.reset();
    return {m, c};
  }

  // The (scalar, base) pairs `evaluate` hands to the loader: constant * gen
  // first, then the terms in insertion order (msm.rs:81-98).
  std::vector<std::pair<Scalar, Point>> pairs(const std::optional<Point>& gen) const {
    std::vector<std::pair<Scalar, Point>> out;
    out.reserve(scalars.size() + 1);
    if (constant) {
      if (!gen) throw Panic("Msm has a constant but no generator was given (reference: unwrap, msm.rs:93)");
      out.emplace_back(*constant, L::ec_point_load_const(*gen));
    }
    for (size_t i = 0; i < scalars.size(); ++i) out.emplace_back(scalars[i], *bases[i]);
    return out;
  }

  // msm.rs:81-98
  Point evaluate(const std::optional<Point>& gen) const {
    auto prs = pairs(gen);
    std::vector<std::pair<const Scalar*, const Point*>> refs;
    refs.reserve(prs.size());
    for (auto& pr : prs) refs.emplace_back(&pr.first, &pr.second);
    return L::multi_scalar_multiplication(refs);
  }

  // msm.rs:100-107
  void scale(const Scalar& f) {
    if (constant) *constant *= f;
    for (auto& s : scalars) s *= f;
  }
  // msm.rs:109-116: equal bases are merged.  The reference scans linearly
  // (`position`), which is quadratic over the (m+1)-term Msm of KzgAs::verify;
  // past a few dozen terms the same merge runs through a hash index.
  void push(const Scalar& s, const Point* b) {
    if (bases.size() < kIndexThreshold) {
      for (size_t i = 0; i < bases.size(); ++i)
        if (*bases[i] == *b) {
          scalars[i] += s;
          return;
        }
    } else {
      if (!index_ || index_->size() != bases.size()) rebuild_index();
      auto range = index_->equal_range(key_of(*b));
      for (auto it = range.first; it != range.second; ++it)
        if (*bases[it->second] == *b) {
          scalars[it->second] += s;
          return;
        }
      index_->emplace(key_of(*b), bases.size());
    }
    scalars.push_back(s);
    bases.push_back(b);
  }
  // msm.rs:118-127
  void extend(const Msm& o) {
    if (o.constant) {
      if (constant) *constant += *o.constant;
      else constant = o.constant;
    }
    for (size_t i = 0; i < o.scalars.size(); ++i) push(o.scalars[i], o.bases[i]);
  }

  // msm.rs:130-226
  Msm operator+(const Msm& o) const {
    Msm r = *this;
    r.extend(o);
    return r;
  }
  Msm& operator+=(const Msm& o) {
    extend(o);
    return *this;
  }
  Msm operator-() const {
    Msm r = *this;
    if (r.constant) r.constant = -*r.constant;
    for (auto& s : r.scalars) s = -s;
    return r;
  }
  Msm operator-(const Msm& o) const { return *this + (-o); }
  Msm& operator-=(const Msm& o) { return *this += (-o); }
  Msm operator*(const Scalar& f) const {
    Msm r = *this;
    r.scale(f);
    return r;
  }
  Msm& operator*=(const Scalar& f) {
    scale(f);
    return *this;
  }
  template <class It>
  static Msm sum(It first, It last) {
    if (first == last) return Msm();
    Msm acc = *first;
    for (++first; first != last; ++first) acc.extend(*first);
    return acc;
  }
  static Msm sum(const std::vector<Msm>& v) { return sum(v.begin(), v.end()); }
};

// `util::msm::multi_scalar_multiplication(&[C::Scalar], &[C]) -> C::Curve`
// (reference snark-verifier/src/util/msm.rs:308-343): the large-MSM free function
// (IPA commit / decide) -> the device Pippenger.  Returns the affine point (the
// reference's callers all `to_affine()` it).  Panics like the reference on a
// length mismatch (`assert_eq!`, msm.rs:309) and on n = 0 (`scalars[0]`, msm.rs:265).
inline G1Affine multi_scalar_multiplication(const std::vector<Fr>& scalars, const std::vector<G1Affine>& bases) {
  if (scalars.size() != bases.size()) throw Panic("multi_scalar_multiplication: scalars.len() != bases.len() (msm.rs:309)");
  if (scalars.empty()) throw Panic("multi_scalar_multiplication of no terms (reference: scalars[0], msm.rs:265)");
  const size_t n = scalars.size();
  // packed into this thread's pinned host buffers (`bn254_host_buffer`; loader.hpp `DeviceScope`): 96 B per term reach the device
  // by DMA, and a repeated call does not fault in 96 B per term of fresh pageable memory first
  DeviceScope lock;
  uint8_t *s = nullptr, *p = nullptr;
  if (SNARKV_DEV(host_buffer)(0, 32 * n, (void**)&s) != SNARKV_OK || SNARKV_DEV(host_buffer)(1, 64 * n, (void**)&p) != SNARKV_OK)
    throw std::runtime_error(std::string("host_buffer: ") + SNARKV_DEV_LAST_ERROR());
  auto pack = [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      scalars[i].to_bytes(&s[32 * i]);
      memcpy(&p[64 * i], bases[i].b, 64);
    }
  };
  if (n >= 8192) {
    const size_t per = 2048, tasks = (n + per - 1) / per;
    parallel_for(tasks, 16, [&](size_t t) { pack(t * per, std::min(n, (t + 1) * per)); }, 1);
  } else {
    pack(0, n);
  }
  G1Affine out;
  int rc = SNARKV_DEV(g1_msm_pippenger)(s, p, n, out.b);
  if (rc != SNARKV_OK) throw std::runtime_error(std::string("g1_msm_pippenger: ") + SNARKV_DEV_LAST_ERROR());
  return out;
}

}  // namespace snarkv_host
