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
// so the device state is process-global here too (the C ABI's default context).
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
class HostPool {
 public:
  static HostPool& get() {
    static HostPool p;
    return p;
  }
  unsigned size() const { return (unsigned)workers_.size(); }
  // runs fn(i) for i in [0, n) on up to `threads` pool workers (the caller blocks); the first exception wins
  template <class F>
  void run(size_t n, unsigned threads, F&& fn) {
    threads = std::min<unsigned>(threads, size());
    if (threads <= 1 || n <= 1 || in_worker()) {  // a task that itself fans out runs its items inline
      for (size_t i = 0; i < n; ++i) fn(i);
      return;
    }
    std::lock_guard<std::mutex> one_job(submit_mu_);  // one job at a time keeps the bookkeeping trivial
    Job job;
    job.n = n;
    job.fn = [&](size_t i) { fn(i); };
    job.slots = threads;
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &job;
      ++generation_;
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return job.finished == job.started && job.next.load() >= n && job.slots_left() == false; });
    job_ = nullptr;
    lk.unlock();
    if (job.err) std::rethrow_exception(job.err);
  }
  // true on the pool's own threads (a task that fans out runs inline; a task must never take the device lock)
  static bool& in_worker() {
    static thread_local bool flag = false;
    return flag;
  }

 private:
  struct Job {
    size_t n = 0;
    std::function<void(size_t)> fn;
    std::atomic<size_t> next{0};
    unsigned slots = 0, started = 0, finished = 0;  // guarded by mu_
    std::exception_ptr err;
    std::atomic<bool> failed{false};
    bool slots_left() const { return started < slots && next.load() < n; }
  };
  HostPool() {
    unsigned hc = std::max(1u, std::thread::hardware_concurrency());
    unsigned k = std::min(64u, hc);
    if (const char* e = getenv("SNARKV_HOST_POOL")) k = (unsigned)std::max(1, std::min(512, atoi(e)));  // tuning knob
    for (unsigned i = 0; i < k; ++i) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  void loop() {
    in_worker() = true;
    uint64_t seen = 0;
    for (;;) {
      Job* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || (job_ && generation_ != seen && job_->slots_left()); });
        if (stop_) return;
        seen = generation_;
        job = job_;
        ++job->started;
      }
      for (;;) {
        size_t i = job->next.fetch_add(1);
        if (i >= job->n || job->failed.load()) break;
        try {
          job->fn(i);
        } catch (...) {
          if (!job->failed.exchange(true)) job->err = std::current_exception();
          job->next.store(job->n);  // nothing more to hand out
          break;
        }
      }
      {
        std::lock_guard<std::mutex> lk(mu_);
        ++job->finished;
      }
      done_cv_.notify_all();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, submit_mu_;
  std::condition_variable cv_, done_cv_;
  Job* job_ = nullptr;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

// `grain`: items per thread below which another thread is not worth waking.
template <class F>
inline void parallel_for(size_t n, unsigned threads, F&& fn, size_t grain = 16) {
  threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(1, n / std::max<size_t>(1, grain)));
  HostPool::get().run(n, threads, std::forward<F>(fn));
}

// The context-free C-ABI entry points share one process-global device context
// (`multi_scalar_multiplication` has no `&self`, loader.rs:108): host threads
// that reach the device through the loader take turns.  The lock also covers the default context's pinned host
// buffers (packing happens under it), and `parallel_for` may be called while holding it: tasks of the pool therefore
// never take this lock (they are host-only parsing and field algebra).
inline std::mutex& device_mutex() {
  static std::mutex m;
  // lock order: device_mutex -> the pool's submit lock (packing fans out under the device lock).  A pool task that took
  // the device lock would close the cycle: refuse it loudly instead of deadlocking some day.
  if (HostPool::in_worker()) throw std::logic_error("device_mutex() from a pool worker: tasks of the host pool must stay host-only");
  return m;
}

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
    std::lock_guard<std::mutex> lock(device_mutex());
    int rc = SNARKV_DEV(g1_msm_naive)(s.data(), p.data(), pairs.size(), out.b);
    if (rc != SNARKV_OK) throw std::runtime_error(std::string("g1_msm_naive: ") + SNARKV_DEV_LAST_ERROR());
    return out;
  }

  // Beyond the reference surface: several deferred MSMs in ONE segmented launch
  // (what makes the GPU worthwhile for the many small MSMs of accumulation).
  static std::vector<G1Affine> multi_scalar_multiplication_batch(
      const std::vector<std::vector<std::pair<Fr, G1Affine>>>& msms) {
    std::vector<uint32_t> offs(1, 0);
    for (auto& m : msms) {
      if (m.empty()) throw Panic("empty MSM in batch (reference: native.rs:69)");
      offs.push_back(offs.back() + (uint32_t)m.size());
    }
    const size_t total = offs.back();
    // the terms are packed straight into the device library's pinned host buffers (the copy to the device is then a
    // DMA, not the runtime's bounce copy of pageable memory); they belong to the default context, so the device lock
    // is taken before packing
    std::lock_guard<std::mutex> lock(device_mutex());
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
    if (total >= 4096) {
      const size_t per = 64;  // MSMs per task
      const size_t tasks = (msms.size() + per - 1) / per;
      parallel_for(tasks, 16, [&](size_t k) { pack(k * per, std::min(msms.size(), (k + 1) * per)); }, 1);
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
