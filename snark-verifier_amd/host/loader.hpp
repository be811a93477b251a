// Mirror of the reference's Loader abstraction for the native (value) loader,
// with the EC work routed to the MI355X through the C ABI.
//
//   reference                                         here
//   `EcPointLoader` / `ScalarLoader` / `Loader`       GpuNativeLoader
//     snark-verifier/src/loader.rs:82-274
//   `NativeLoader` (unit struct, global LOADER)       GpuNativeLoader (static fns,
//     snark-verifier/src/loader/native.rs:11-93         process-global device context)
//   `LoadedEcPoint = C`, `LoadedScalar = F`           G1Affine (64 canonical bytes), Fr
//
// `multi_scalar_multiplication` has no `&self` in the reference (loader.rs:108),
// so the device state is process-global here too (the C ABI's pool of default contexts).
#pragma once
#include <cstdint>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/snarkv_amd.h"
#include "fr.hpp"

// The context-free device entry points the loader binds: libsnarkv_amd.so (bn254_*) or, for the pasta
// flavour of the mirror (-DSNARKV_HOST_PALLAS: host/test_driver_pallas.cpp), libsnarkv_pallas.so (pallas_*).
#if defined(SNARKV_HOST_PALLAS)
#include "../../include/snarkv_pallas.h"
#define SNARKV_DEV(name) pallas_##name
#define SNARKV_DEV_LAST_ERROR snarkv_pallas_last_error
#define SNARKV_DEV_IPA_DK_DESTROY snarkv_pallas_ipa_dk_destroy
#else
#define SNARKV_DEV(name) bn254_##name
#define SNARKV_DEV_LAST_ERROR snarkv_last_error
#define SNARKV_DEV_IPA_DK_DESTROY snarkv_ipa_dk_destroy
#endif

namespace snarkv_host {

// `snark_verifier::Error` (reference snark-verifier/src/lib.rs:18-28)
struct Error {
  enum Kind { None = 0, InvalidInstances, InvalidProtocol, AssertionFailure, Transcript } kind = None;
  std::string msg;
  bool ok() const { return kind == None; }
  static Error assertion(const std::string& m) { return Error{AssertionFailure, m}; }
};

// The reference PANICS in these spots (unwrap/assert); the mirror throws.
struct Panic : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// `G1Affine` as it crosses the boundary: x || y canonical LE; identity = zeros.
struct G1Affine {
  uint8_t b[64];
  G1Affine() { memset(b, 0, 64); }
  static G1Affine from_bytes(const uint8_t* p) {
    G1Affine r;
    memcpy(r.b, p, 64);
    return r;
  }
#if !defined(SNARKV_HOST_PALLAS)
  static G1Affine generator() {  // bn256 G1: (1, 2)
    G1Affine r;
    r.b[0] = 1;
    r.b[32] = 2;
    return r;
  }
#endif
  static G1Affine identity() { return G1Affine(); }
  bool is_identity() const {
    for (int i = 0; i < 64; ++i)
      if (b[i]) return false;
    return true;
  }
  bool operator==(const G1Affine& o) const { return memcmp(b, o.b, 64) == 0; }
};

struct G2Affine {
  uint8_t b[128];
  static G2Affine from_bytes(const uint8_t* p) {
    G2Affine r;
    memcpy(r.b, p, 128);
    return r;
  }
};

// Proofs are independent: the host front half (transcript hashing, expression
// evaluation) is spread over host threads.  A persistent pool: starting a thread costs
// about as much as the host work of one Keccak-transcript proof, and an aggregation
// makes several passes.
//
// One job at a time.  The job lives IN the pool (stable memory), identified by a generation number; a worker joins it by
// a compare-and-swap on `state_` = generation << 32 | closed << 31 | workers inside, so joining, leaving and the caller's
// "everybody out" need no lock, and a worker that arrives late finds another generation (or the closed bit) and never
// touches the job's fields.  Workers spin briefly for the next generation before they sleep on the condition variable:
// an aggregation runs five or six parallel passes of a few hundred microseconds back to back, and waking 64 sleepers
// through one mutex cost each pass a good part of its own duration.  The caller works on the items too.
class HostPool {
 public:
  static HostPool& get() {
    static HostPool p;
    return p;
  }
  unsigned size() const { return (unsigned)workers_.size(); }
  // consecutive items a worker takes per claim when `run(n, threads, ..)` hands them out (a caller that orders its items
  // for the claims: aggregation.hpp)
  static size_t claim_size(size_t n, unsigned threads) { return std::max<size_t>(1, n / ((size_t)std::max(1u, threads) * 8)); }
  // runs fn(i) for i in [0, n) on up to `threads` threads, the caller among them (it returns when all are done); the
  // first exception wins
  template <class F>
  void run(size_t n, unsigned threads, F&& fn) {
    threads = std::min<unsigned>(threads, size() + 1);
    if (threads <= 1 || n <= 1 || in_worker()) {  // a task that itself fans out runs its items inline
      for (size_t i = 0; i < n; ++i) fn(i);
      return;
    }
    std::lock_guard<std::mutex> one_job(submit_mu_);
    // (the previous job is closed and empty: nobody reads these fields now)
    n_.store(n, std::memory_order_relaxed);
    chunk_.store(claim_size(n, threads), std::memory_order_relaxed);  // items per claim: 8 claims per thread
    fn_ = [&](size_t i) { fn(i); };
    next_.store(0, std::memory_order_relaxed);
    max_workers_.store(threads - 1, std::memory_order_relaxed);
    failed_.store(false, std::memory_order_relaxed);
    err_ = nullptr;
    const uint64_t gen = (state_.load(std::memory_order_relaxed) >> 32) + 1;
    state_.store(gen << 32, std::memory_order_seq_cst);  // open
    {
      std::lock_guard<std::mutex> lk(mu_);  // a worker between its predicate and its wait holds mu_: no lost wake-up
    }
    cv_.notify_all();
    in_worker() = true;  // the caller's share: its tasks are pool tasks like any other (no device scope, no nested fan-out)
    work();
    in_worker() = false;
    // close the job, then wait until the workers inside have left
    uint64_t st = state_.load(std::memory_order_acquire);
    while (!state_.compare_exchange_weak(st, st | kClosed, std::memory_order_acq_rel)) {
    }
    for (unsigned spins = 0; (state_.load(std::memory_order_acquire) & kCountMask) != 0; ++spins) {
      if (spins < 4096) cpu_relax();
      else std::this_thread::yield();
    }
    fn_ = nullptr;
    if (err_) std::rethrow_exception(err_);
  }
  // true on the pool's own threads and on a caller while it works on its job (a task that fans out runs inline; a task
  // must never open a device scope)
  static bool& in_worker() {
    static thread_local bool flag = false;
    return flag;
  }

 private:
  static constexpr uint64_t kClosed = 1ull << 31, kCountMask = kClosed - 1;
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
  }
  HostPool() {
    unsigned hc = std::max(1u, std::thread::hardware_concurrency());
    unsigned k = std::min(64u, hc);
    if (const char* e = getenv("SNARKV_HOST_POOL")) k = (unsigned)std::max(1, std::min(512, atoi(e)));  // tuning knob
    for (unsigned i = 0; i < k; ++i) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_.store(true);
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  // runs of items until none are left (or a task threw)
  void work() {
    const size_t n = n_.load(std::memory_order_relaxed), chunk = chunk_.load(std::memory_order_relaxed);
    for (;;) {
      const size_t i0 = next_.fetch_add(chunk, std::memory_order_relaxed);
      if (i0 >= n || failed_.load(std::memory_order_relaxed)) break;
      try {
        for (size_t i = i0, e = std::min(n, i0 + chunk); i < e; ++i) fn_(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(err_mu_);
        if (!failed_.exchange(true)) err_ = std::current_exception();
        next_.store(n, std::memory_order_relaxed);  // nothing more to hand out
        break;
      }
    }
  }
  void loop() {
    in_worker() = true;
    uint64_t seen = 0;  // generation this worker has dealt with
    for (;;) {
      // wait for another generation: spin first (the next pass of an aggregation is usually microseconds away)
      uint64_t st = state_.load(std::memory_order_acquire);
      for (unsigned spins = 0; (st >> 32) == seen && spins < 5000 && !stop_.load(std::memory_order_relaxed); ++spins) {
        cpu_relax();
        st = state_.load(std::memory_order_acquire);
      }
      if ((st >> 32) == seen) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_.load() || (state_.load(std::memory_order_acquire) >> 32) != seen; });
        st = state_.load(std::memory_order_acquire);
      }
      if (stop_.load()) return;
      seen = st >> 32;
      // join unless the job is closed, full, or already another one
      bool joined = false;
      while ((st >> 32) == seen && !(st & kClosed) && (st & kCountMask) < max_workers_.load(std::memory_order_relaxed) &&
             next_.load(std::memory_order_relaxed) < n_.load(std::memory_order_relaxed)) {
        if (state_.compare_exchange_weak(st, st + 1, std::memory_order_acq_rel)) {
          joined = true;
          break;
        }
      }
      if (!joined) continue;
      work();
      state_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, submit_mu_, err_mu_;
  std::condition_variable cv_;
  std::atomic<uint64_t> state_{0};
  std::atomic<bool> stop_{false}, failed_{false};
  // the job (valid for the generation in state_ while it is open or a worker is inside)
  std::atomic<size_t> n_{0}, chunk_{1};
  std::function<void(size_t)> fn_;
  std::atomic<size_t> next_{0};
  std::atomic<unsigned> max_workers_{0};
  std::exception_ptr err_;
};

// `grain`: items per thread below which another thread is not worth waking.
template <class F>
inline void parallel_for(size_t n, unsigned threads, F&& fn, size_t grain = 16) {
  threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(1, n / std::max<size_t>(1, grain)));
  HostPool::get().run(n, threads, std::forward<F>(fn));
}

// A scope in which the calling thread talks to the device through the context-free C-ABI entry points
// (`multi_scalar_multiplication` has no `&self`, loader.rs:108).
//   bn254 flavour: the device library keeps a POOL of default contexts and hands one out per call, and the pinned host
//     buffers of `bn254_host_buffer` belong to the calling thread -- so host threads reach the device side by side and
//     the scope is NOT a lock (round 4: one process-wide mutex = one job at a time through the trait boundary).  What it
//     does: pin THIS thread's bn254_* calls to the wire form (flags 0) whatever the application set with
//     `bn254_set_flags` -- the mirror hands canonical bytes to the device -- and restore on the way out.
//   pasta flavour: libsnarkv_pallas.so still has one default context; there the scope is the process-wide lock.
// Either way tasks of the host pool stay host-only: a worker that blocked on the device would stall every job that
// shares the pool, so a scope opened from a worker is refused loudly.
struct DeviceScope {
  DeviceScope() {
    if (HostPool::in_worker()) throw std::logic_error("DeviceScope from a pool worker: tasks of the host pool must stay host-only");
#if defined(SNARKV_HOST_PALLAS)
    mu().lock();
#else
    saved_ = bn254_set_thread_flags(0);
#endif
  }
  ~DeviceScope() {
#if defined(SNARKV_HOST_PALLAS)
    mu().unlock();
#else
    bn254_set_thread_flags(saved_);
#endif
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;

 private:
#if defined(SNARKV_HOST_PALLAS)
  static std::mutex& mu() {
    static std::mutex m;
    return m;
  }
#else
  int64_t saved_ = -1;
#endif
};

struct GpuNativeLoader {
  using LoadedScalar = Fr;
  using LoadedEcPoint = G1Affine;

  // ---- ScalarLoader (loader.rs:116-263)
  static Fr load_const(const Fr& v) { return v; }
  static Fr load_zero() { return Fr::zero(); }
  static Fr load_one() { return Fr::one(); }
  static Error assert_eq(const std::string& annotation, const Fr& l, const Fr& r) {
    return l == r ? Error{} : Error::assertion(annotation);
  }
  // default `batch_invert`: per-element invert, zeros stay (loader.rs:255-262)
  static void batch_invert(std::vector<Fr*>& values) {
    for (Fr* v : values) {
      Fr inv;
      if (v->invert(&inv)) *v = inv;
    }
  }
  static Fr sum_products(const std::vector<std::pair<Fr, Fr>>& terms) {
    Fr acc;
    for (auto& t : terms) acc += t.first * t.second;
    return acc;
  }

  // ---- EcPointLoader (loader.rs:82-113)
  static G1Affine ec_point_load_const(const G1Affine& v) { return v; }
  static Error ec_point_assert_eq(const std::string& annotation, const G1Affine& l, const G1Affine& r) {
    return l == r ? Error{} : Error::assertion(annotation);
  }

  // THE drop-in point: `NativeLoader::multi_scalar_multiplication`
  // (reference snark-verifier/src/loader/native.rs:61-71) on the MI355X.
  static G1Affine multi_scalar_multiplication(const std::vector<std::pair<const Fr*, const G1Affine*>>& pairs) {
    if (pairs.empty()) throw Panic("multi_scalar_multiplication of no pairs (reference: reduce().unwrap(), native.rs:69)");
    std::vector<uint8_t> s(32 * pairs.size()), p(64 * pairs.size());
    for (size_t i = 0; i < pairs.size(); ++i) {
      pairs[i].first->to_bytes(&s[32 * i]);
      memcpy(&p[64 * i], pairs[i].second->b, 64);
    }
    G1Affine out;
    DeviceScope lock;
    int rc = SNARKV_DEV(g1_msm_naive)(s.data(), p.data(), pairs.size(), out.b);
    if (rc != SNARKV_OK) throw std::runtime_error(std::string("g1_msm_naive: ") + SNARKV_DEV_LAST_ERROR());
    return out;
  }

  // Beyond the reference surface: several deferred MSMs in ONE segmented launch
  // (what makes the GPU worthwhile for the many small MSMs of accumulation).
  // `use_pool`: false = pack on the calling thread whatever the size (a caller whose host pool is busy with another pass:
  // the pool runs one job at a time, and waiting for it would serialise the two -- aggregation.hpp's device threads)
  static std::vector<G1Affine> multi_scalar_multiplication_batch(
      const std::vector<std::vector<std::pair<Fr, G1Affine>>>& msms, bool use_pool = true) {
    std::vector<uint32_t> offs(1, 0);
    for (auto& m : msms) {
      if (m.empty()) throw Panic("empty MSM in batch (reference: native.rs:69)");
      offs.push_back(offs.back() + (uint32_t)m.size());
    }
    const size_t total = offs.back();
    // the terms are packed straight into the device library's pinned host buffers (the copy to the device is then a
    // DMA, not the runtime's bounce copy of pageable memory); they belong to THIS thread, so the device scope
    // is taken before packing
    DeviceScope lock;
    uint8_t *s = nullptr, *p = nullptr;
    if (SNARKV_DEV(host_buffer)(0, 32 * total, (void**)&s) != SNARKV_OK || SNARKV_DEV(host_buffer)(1, 64 * total, (void**)&p) != SNARKV_OK)
      throw std::runtime_error(std::string("host_buffer: ") + SNARKV_DEV_LAST_ERROR());
    auto pack = [&](size_t lo, size_t hi) {  // Montgomery -> canonical bytes is one field product per scalar
      for (size_t k = lo; k < hi; ++k) {
        size_t o = offs[k];
        for (auto& t : msms[k]) {
          t.first.to_bytes(&s[32 * o]);
          memcpy(&p[64 * o], t.second.b, 64);
          ++o;
        }
      }
    };
    if (total >= 4096 && use_pool) {
      const size_t per = 16;  // MSMs per task (a few hundred terms: ~10 us of conversions and copies)
      const size_t tasks = (msms.size() + per - 1) / per;
      parallel_for(tasks, 64, [&](size_t k) { pack(k * per, std::min(msms.size(), (k + 1) * per)); }, 1);
    } else {
      pack(0, msms.size());
    }
    std::vector<G1Affine> out(msms.size());
    int rc = SNARKV_DEV(g1_msm_batched)(s, p, offs.data(), msms.size(), out.empty() ? nullptr : out[0].b);
    if (rc != SNARKV_OK) throw std::runtime_error(std::string("g1_msm_batched: ") + SNARKV_DEV_LAST_ERROR());
    return out;
  }
};

}  // namespace snarkv_host
