// C ABI glue (include/snarkv_amd.h): context, staging, error reporting.
// No arithmetic happens on the host; every entry point stages bytes to HBM,
// enqueues the HIP kernels and copies the (tiny) result back.
#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include "ctx.hpp"

#include "ctx_impl.inc"

namespace snarkv {

// THE decision "does an n-point MSM run as the chunk pipeline over shared bucket grids?" -- one rule for the single
// call (launch_msm_pippenger_auto), the batch (launch_msm_pippenger_many hands such jobs to the single call) and
// snarkv_g1_msm_launch_points (what bench.py divides its per-launch roofline by).  *chunk = points per chunk.
//   SNARKV_PIP_SPLIT   0 never, 1 (default) from three chunks, 2 from two
// An explicit window size or a lane context (a worker of a pipeline already) keeps the single launch.
bool pip_chunk_pipeline(size_t n, int window_bits, bool is_lane, size_t* chunk) {
  const size_t c = (size_t)1 << 20;  // chunk size: the 2^20-point MSM's window geometry, point table inside the Infinity Cache
  const char* e = getenv("SNARKV_PIP_SPLIT");
  const int mode = e ? atoi(e) : 1;
  const size_t min_chunks = mode == 2 ? 2 : 3;
  if (chunk) *chunk = c;
  return mode != 0 && (n + c - 1) / c >= min_chunks && window_bits == 0 && !is_lane;
}

static int stage_in(snarkv_ctx* ctx, int slot, const void* host, size_t bytes, void** d) {
  SNARKV_TRY(ctx_reserve(ctx, slot, bytes, d));
  SNARKV_HIP(hipMemcpyAsync(*d, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return SNARKV_OK;
}

static int fetch_out(snarkv_ctx* ctx, const void* d, void* host, size_t bytes) {
  SNARKV_HIP(hipMemcpyAsync(host, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  return SNARKV_OK;
}

// HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams sharing a queue serialise.
// A server that keeps many latency-bound aggregation jobs in flight on contexts of their own wants 16 (2.5x the proofs/s
// of 64-proof jobs, profiles/r03_agg_hw_queues.txt).  That is a process-wide runtime setting read at the runtime's first
// call, so the library does NOT touch it when it is loaded: the caller exports GPU_MAX_HW_QUEUES=16 before its first HIP
// call (the Python package does so on import unless the variable is set, INTEGRATION.md shows the Rust line).

// ---- the context-free entry points' contexts ------------------------------------------------------------------------
// `EcPointLoader::multi_scalar_multiplication` has no `&self` (loader.rs:108), so a trait-bound Rust caller can only reach
// the `bn254_*` forms -- and round 4 gave them ONE context behind one mutex: a rayon-parallel caller got one job at a
// time, a sixteenth of what sixteen explicit contexts deliver.  Now they draw from a POOL of default
// contexts (a stream + scratch each), checked out per call:
//   * size: SNARKV_DEFAULT_CONTEXTS, else GPU_MAX_HW_QUEUES (the hardware queues the process's streams are spread over),
//     else 4 (the runtime's default queue count); contexts are created on demand, never destroyed;
//   * affinity: a thread comes back to the context it used last when that one is free (warm scratch), else any free
//     one, else a new one while the pool may grow, else it waits;
//   * nesting: bn254_kzg_decide -> bn254_kzg_decide_batch runs on the context the outer call holds;
//   * flags: the process default (bn254_set_flags) or the calling thread's override (bn254_set_thread_flags) is applied
//     to the context at check-out, and so is the throughput hint (on while eight or more contexts are checked out);
//   * pinned host buffers (bn254_host_buffer) belong to the calling THREAD, not to a context: what a thread packs is
//     its own until it asks again, whatever context its next call lands on.
struct DefaultPool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<snarkv_ctx*> ctx;
  std::vector<char> busy;
  int cap = 0;
  std::atomic<uint32_t> flags{0};
};
static DefaultPool g_pool;
static thread_local int tl_slot = -1, tl_depth = 0;
static thread_local snarkv_ctx* tl_held = nullptr;
static thread_local int64_t tl_flags = -1;  // -1: the process default

static int pool_cap() {
  for (const char* name : {"SNARKV_DEFAULT_CONTEXTS", "GPU_MAX_HW_QUEUES"})
    if (const char* e = getenv(name)) {
      int v = atoi(e);
      if (v > 0) return std::min(v, 64);
    }
  return 4;
}

struct DefaultLease {
  snarkv_ctx* c = nullptr;
  int rc = SNARKV_OK;
  DefaultLease() {
    if (tl_depth > 0) {  // nested context-free call: the context the outer call holds
      c = tl_held;
      ++tl_depth;
      return;
    }
    std::unique_lock<std::mutex> lk(g_pool.mu);
    if (g_pool.cap == 0) g_pool.cap = pool_cap();
    int slot = -1;
    for (;;) {
      if (tl_slot >= 0 && tl_slot < (int)g_pool.ctx.size() && !g_pool.busy[tl_slot]) slot = tl_slot;
      for (int i = 0; slot < 0 && i < (int)g_pool.ctx.size(); ++i)
        if (!g_pool.busy[i]) slot = i;
      if (slot >= 0) break;
      if ((int)g_pool.ctx.size() < g_pool.cap) {
        snarkv_ctx* fresh = nullptr;
        rc = snarkv_ctx_create(0, nullptr, &fresh);  // (under the pool lock: at most `cap` times per process)
        if (rc < 0) return;
        g_pool.ctx.push_back(fresh);
        g_pool.busy.push_back(0);
        slot = (int)g_pool.ctx.size() - 1;
        break;
      }
      g_pool.cv.wait(lk);
    }
    g_pool.busy[slot] = 1;
    c = g_pool.ctx[slot];
    int in_use = 0;
    for (char b : g_pool.busy) in_use += b ? 1 : 0;
    lk.unlock();
    // the throughput hint, by what the pool sees: with eight or more calls in flight the GPU is shared and the forms that
    // do less work on a longer chain pay (1 024-proof jobs: 0.83 -> 0.75 ms per job at 8 in flight, 0.70 -> 0.49 at 16);
    // below that they lose (1.35 -> 2.06 at 2, 1.04 -> 1.22 at 4: tools/aggregate_inflight.py --hint-from)
    c->throughput_mode = in_use >= 8;
    c->throughput_peers = in_use;  // the pool KNOWS how many calls share the GPU (not a fixed sixteenth)
    tl_slot = slot;
    tl_held = c;
    tl_depth = 1;
    const uint32_t f = tl_flags >= 0 ? (uint32_t)tl_flags : g_pool.flags.load();
    c->flags = f;
    c->mont = (f & SNARKV_FLAG_MONTGOMERY) != 0;
  }
  ~DefaultLease() {
    if (!c) return;
    if (--tl_depth > 0) return;
    tl_held = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_pool.mu);
      g_pool.busy[tl_slot] = 0;
    }
    g_pool.cv.notify_one();
  }
};
#define SNARKV_DEFAULT_LEASE(c)   \
  DefaultLease _lease;            \
  if (_lease.rc < 0) return _lease.rc; \
  snarkv_ctx* c = _lease.c

// pinned host buffers of the context-free callers, one set per THREAD (slots reused when a thread ends; the memory is
// never returned: like the contexts, it lives as long as the process)
struct ThreadHostBufs {
  void* buf[SNARKV_HOST_BUFFERS] = {};
  size_t cap[SNARKV_HOST_BUFFERS] = {};
};
static std::mutex g_hb_mu;
static std::vector<ThreadHostBufs*> g_hb_free;
struct ThreadHostBufsRef {
  ThreadHostBufs* p = nullptr;
  ~ThreadHostBufsRef() {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_hb_mu);
    g_hb_free.push_back(p);  // no HIP call at thread exit: the set goes to the next thread that asks
  }
};
static thread_local ThreadHostBufsRef tl_hb;

// line tables of the last DK_CACHE deciding keys of bn254_kzg_decide[_batch] (file scope: bn254_shutdown releases them)
enum { DK_CACHE = 4 };
struct DkCacheEntry {
  std::shared_ptr<snarkv_dk> dk;
  uint8_t tag[321];
  uint64_t used;
};
static std::mutex g_dk_cache_mu;
static DkCacheEntry g_dk_cache[DK_CACHE];
static uint64_t g_dk_tick = 0;

}  // namespace snarkv

using namespace snarkv;

extern "C" {

int snarkv_set_stage_timing(snarkv_ctx* ctx, int enabled) {
  if (!ctx) return SNARKV_ERR_ARG;
  ctx->stage_timing = enabled != 0;
  return SNARKV_OK;
}

int snarkv_ctx_set_flags(snarkv_ctx* ctx, uint32_t flags) {
  if (!ctx || (flags & ~(SNARKV_FLAG_VALIDATE | SNARKV_FLAG_MONTGOMERY))) return SNARKV_ERR_ARG;
  ctx->flags = flags;
  ctx->mont = (flags & SNARKV_FLAG_MONTGOMERY) != 0;
  return SNARKV_OK;
}

uint32_t snarkv_ctx_get_flags(const snarkv_ctx* ctx) { return ctx ? ctx->flags : 0u; }

int snarkv_ctx_set_throughput_hint(snarkv_ctx* ctx, int enabled) {
  if (!ctx) return SNARKV_ERR_ARG;
  ctx->throughput_mode = enabled != 0;
  ctx->throughput_peers = enabled > 1 ? enabled : 0;  // n >= 2: that many contexts in flight; 1: unknown (16 assumed)
  return SNARKV_OK;
}

int snarkv_g1_msm_launch_points_ex(size_t n, int window_bits, size_t* per_launch) {
  if (!per_launch) return SNARKV_ERR_ARG;
  size_t chunk = 0;
  *per_launch = snarkv::pip_chunk_pipeline(n, window_bits, false, &chunk) ? chunk : n;
  return SNARKV_OK;
}

int snarkv_g1_msm_launch_points(size_t n, size_t* per_launch) { return snarkv_g1_msm_launch_points_ex(n, 0, per_launch); }

int snarkv_get_stage_timing(snarkv_ctx* ctx, float ms[SNARKV_PIP_STAGES]) {
  if (!ctx || !ctx->ev_ready) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->last_many_jobs > 0) {
    // batch (snarkv_g1_msm_pippenger_many_dev): [0] the whole call; [4] / [5] ONE k_accumulate / combine launch of its
    // last round, averaged over the jobs' own events (launches of up to three jobs overlap); the rest is not broken down
    for (int i = 0; i < SNARKV_PIP_STAGES; ++i) ms[i] = 0.f;
    SNARKV_HIP(hipEventElapsedTime(&ms[0], ctx->ev[0], ctx->ev[SNARKV_PIP_STAGES - 1]));
    const float jobs = (float)ctx->last_many_jobs;
    for (int j = 0; j < ctx->last_many_jobs; ++j) {
      snarkv_ctx* job = ctx->jobs[j];
      if (!job->ev_ready) continue;
      float t = 0.f;
      SNARKV_HIP(hipEventElapsedTime(&t, job->ev[3], job->ev[4]));
      ms[4] += t / jobs;
      SNARKV_HIP(hipEventElapsedTime(&t, job->ev[4], job->ev[5]));
      ms[5] += t / jobs;
    }
    return SNARKV_OK;
  }
  if (ctx->last_split_workers > 0) {
    // chunk pipeline: total = the whole MSM on this stream; stages 1..5 = ONE 2^20-point chunk (the last one of every
    // worker lane, averaged): what a single launch of each kernel took; the shared tail is not broken down
    SNARKV_HIP(hipEventElapsedTime(&ms[0], ctx->ev[0], ctx->ev[SNARKV_PIP_STAGES - 1]));
    for (int i = 1; i < SNARKV_PIP_STAGES; ++i) ms[i] = 0.f;
    int used = 0;
    for (int w = 0; w < ctx->last_split_workers; ++w) {
      snarkv_ctx* lane = ctx->sub[w];
      if (!lane || !lane->ev_ready) continue;
      SNARKV_HIP(hipStreamSynchronize(lane->stream));
      for (int i = 1; i <= 5; ++i) {
        float t = 0.f;
        SNARKV_HIP(hipEventElapsedTime(&t, lane->ev[i - 1], lane->ev[i]));
        ms[i] += t;
      }
      ++used;
    }
    for (int i = 1; i <= 5 && used; ++i) ms[i] /= (float)used;
    return SNARKV_OK;
  }
  SNARKV_HIP(hipEventElapsedTime(&ms[0], ctx->ev[0], ctx->ev[SNARKV_PIP_STAGES - 1]));
  for (int i = 1; i < SNARKV_PIP_STAGES; ++i) SNARKV_HIP(hipEventElapsedTime(&ms[i], ctx->ev[i - 1], ctx->ev[i]));
  return SNARKV_OK;
}

static int check_validate(snarkv_ctx* ctx, const void* d_s, const void* d_p, size_t n, uint32_t flags) {
  if (!((flags | ctx->flags) & SNARKV_FLAG_VALIDATE)) return SNARKV_OK;
  int bad = 0;
  SNARKV_TRY(launch_validate(ctx, d_s, d_p, n, &bad));
  if (bad) {
    set_last_error("%d of %zu inputs are non-canonical or off-curve", bad, n);
    return SNARKV_ERR_ENCODING;
  }
  return SNARKV_OK;
}

int snarkv_g1_msm_batched(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64,
                          const uint32_t* offsets, size_t n_msm, uint32_t flags, uint8_t* out) {
  if (!ctx || !scalars32 || !points64 || !offsets || !out) return SNARKV_ERR_ARG;
  if (n_msm == 0) return SNARKV_ERR_EMPTY;
  if (offsets[0] != 0) return SNARKV_ERR_LENGTH;
  for (size_t k = 0; k < n_msm; ++k) {
    if (offsets[k + 1] < offsets[k]) return SNARKV_ERR_LENGTH;
    if (offsets[k + 1] == offsets[k]) return SNARKV_ERR_EMPTY;  // reference panics: native.rs:69
  }
  size_t n = offsets[n_msm];
  SNARKV_HIP(hipSetDevice(ctx->device));
  SNARKV_CALL_FLAGS(ctx, flags);
  void *d_s, *d_p, *d_o, *d_out;
  SNARKV_TRY(stage_in(ctx, SLOT_IN_SCALARS, scalars32, n * 32, &d_s));
  SNARKV_TRY(stage_in(ctx, SLOT_IN_POINTS, points64, n * 64, &d_p));
  SNARKV_TRY(stage_in(ctx, SLOT_IN_OFFSETS, offsets, (n_msm + 1) * 4, &d_o));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, n_msm * 64, &d_out));
  SNARKV_TRY(check_validate(ctx, d_s, d_p, n, flags));
  SNARKV_TRY(launch_msm_batched(ctx, d_s, d_p, d_o, n_msm, n, d_out));
  return fetch_out(ctx, d_out, out, n_msm * 64);
}

int snarkv_g1_decompress(snarkv_ctx* ctx, const uint8_t* in32, size_t n, uint8_t* out64, uint8_t* ok) {
  if (!ctx || (n && (!in32 || !out64 || !ok))) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_OK;
  if (n > 0xFFFFFFFFull) return SNARKV_ERR_LENGTH;
  SNARKV_HIP(hipSetDevice(ctx->device));
  void *d_in, *d_out;
  SNARKV_TRY(stage_in(ctx, SLOT_IN_POINTS, in32, n * 32, &d_in));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, n * 64 + n, &d_out));  // the points, then one validity byte each
  SNARKV_TRY(launch_g1_decompress(ctx, d_in, n, d_out, (uint8_t*)d_out + n * 64));
  SNARKV_HIP(hipMemcpyAsync(ok, (const uint8_t*)d_out + n * 64, n, hipMemcpyDeviceToHost, ctx->stream));
  return fetch_out(ctx, d_out, out64, n * 64);
}

int snarkv_g1_msm_naive(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                        uint32_t flags, uint8_t out64[64]) {
  if (n == 0) return SNARKV_ERR_EMPTY;
  if (n > 0xFFFFFFFFull) return SNARKV_ERR_LENGTH;
  uint32_t offsets[2] = {0, (uint32_t)n};
  return snarkv_g1_msm_batched(ctx, scalars32, points64, offsets, 1, flags, out64);
}

int snarkv_g1_msm_batched_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64,
                              const void* d_offsets, size_t n_msm, size_t n_terms, void* d_out) {
  if (!ctx || !d_scalars32 || !d_points64 || !d_offsets || !d_out) return SNARKV_ERR_ARG;
  if (n_msm == 0 || n_terms == 0) return SNARKV_ERR_EMPTY;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_msm_batched(ctx, d_scalars32, d_points64, d_offsets, n_msm, n_terms, d_out);
}

}  // extern "C"

// LARGE MSMs as a chunk pipeline over ONE bucket grid.
//
// Beyond ~2^21 points the single-launch Pippenger degrades: the Montgomery point table (64 B x 2n) outgrows the
// 256 MiB Infinity Cache, so every bucket-accumulate gather goes to HBM (k_accumulate +11 % per point at 2^24), the
// level-1 partition scatters 8-byte entries into thousands of streams (k_sort_scatter_staged: 3.5x write amplification), and
// level-2 slices no longer fit LDS.  MSM is linear (the reference's own chunking, util/msm.rs:311-336), so n points are
// cut into 2^20-point chunks that all use the window size of a 2^20-point MSM; every chunk runs the efficient small-n
// stages (prepare, partition, sort, bucket accumulate, combine) on one of three worker lanes (private sub-contexts: one
// HIP stream + scratch each) and ADDS its bucket sums into its worker's grid (windows x 2^(c-1) XYZZ points, 36 MiB).
// Chunks on different lanes overlap -- the memory-bound partition of one under the VALU-bound accumulation of another --
// and the latency-bound tail (bucket reduce, 2^(cw) shift chains, to_affine: 0.65 ms) is paid ONCE on the sum of the
// three grids instead of once per chunk.  Same group element, same bytes as the single launch.
// Measured (MI355X, one MSM at a time): 2^22 / 2^24 points, see DESIGN.md section 4.
namespace snarkv {
int launch_msm_pippenger_auto(snarkv_ctx* ctx, const void* d_s, const void* d_p, size_t n, int window_bits, void* d_out,
                              bool partial_out);
}
static int pippenger_maybe_split(snarkv_ctx* ctx, const void* d_s, const void* d_p, size_t n, int window_bits,
                                 void* d_out, bool partial_out) {
  return launch_msm_pippenger_auto(ctx, d_s, d_p, n, window_bits, d_out, partial_out);
}
int snarkv::launch_msm_pippenger_auto(snarkv_ctx* ctx, const void* d_s, const void* d_p, size_t n, int window_bits,
                                      void* d_out, bool partial_out) {
  size_t kChunk = 0;
  const bool split = pip_chunk_pipeline(n, window_bits, ctx->is_lane, &kChunk);
  const size_t chunks = (n + kChunk - 1) / kChunk;
  ctx->last_split_workers = 0;
  ctx->last_many_jobs = 0;
  if (!split) return launch_msm_pippenger(ctx, d_s, d_p, n, window_bits, d_out, partial_out);
  SNARKV_TRY(ctx_lanes(ctx));
  const bool tm = ctx->stage_timing;  // per-stage events: on the worker lanes (their LAST chunk); total on this stream
  if (tm && !ctx->ev_ready) {
    for (int i = 0; i <= SNARKV_PIP_STAGES; ++i) SNARKV_HIP(hipEventCreate(&ctx->ev[i]));
    ctx->ev_ready = true;
  }
  if (tm) SNARKV_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
  uint32_t c = 0, windows = 0, bpw = 0;
  SNARKV_TRY(pip_geometry(kChunk, 0, &c, &windows, &bpw));
  const size_t nb = (size_t)windows * bpw, grid_bytes = nb * SNARKV_G1_PARTIAL_BYTES;
  const int kWorkers = 2;  // 2 vs 3 measured level (2^24: 24.7 vs 25.3 ms); two keep the footprint at ~2 GiB
  void *grid[3], *tmp[3];
  bool started[3] = {false, false, false};
  for (int w = 0; w < kWorkers; ++w) {
    SNARKV_TRY(ctx_reserve(ctx->sub[w], SLOT_MGPU_GRID, grid_bytes, &grid[w]));
    SNARKV_TRY(ctx_reserve(ctx->sub[w], SLOT_MGPU_RECV, grid_bytes, &tmp[w]));
  }
  // inputs may still be in flight on the caller's stream
  SNARKV_TRY(ctx_lanes_fork(ctx));
  for (size_t k = 0; k < chunks; ++k) {
    size_t lo = k * kChunk, len = std::min(kChunk, n - lo);
    int w = (int)(k % kWorkers);
    snarkv_ctx* lane = ctx->sub[w];
    lane->mont = ctx->mont;
    lane->stage_timing = tm;
    // the chunk's bucket sums (sanitised XYZZ, zero = identity): straight into the worker's grid the first time, added to it after
    SNARKV_TRY(launch_msm_pippenger(lane, (const char*)d_s + 32 * lo, (const char*)d_p + 64 * lo, len, (int)c, nullptr, false,
                                    started[w] ? tmp[w] : grid[w]));
    if (started[w]) SNARKV_TRY(launch_buckets_add(lane, grid[w], tmp[w], nb));
    started[w] = true;
  }
  for (int w = 0; w < kWorkers; ++w) ctx->sub[w]->stage_timing = false;
  SNARKV_TRY(ctx_lanes_join(ctx));
  for (int w = 1; w < kWorkers; ++w)
    if (started[w]) SNARKV_TRY(launch_buckets_add(ctx, grid[0], grid[w], nb));
  void* d_part;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_SPLIT_PARTIALS, SNARKV_G1_PARTIAL_BYTES, &d_part));
  ctx->last_split_workers = kWorkers;
  int rc = SNARKV_OK;
  if (partial_out) {
    rc = launch_buckets_reduce(ctx, grid[0], c, 0, windows, d_out);
  } else {
    rc = launch_buckets_reduce(ctx, grid[0], c, 0, windows, d_part);
    if (rc == SNARKV_OK) rc = launch_fold_partials(ctx, d_part, 1, d_out, false);
  }
  if (tm && rc == SNARKV_OK) SNARKV_HIP(hipEventRecord(ctx->ev[SNARKV_PIP_STAGES - 1], ctx->stream));
  return rc;
}

// ---- many MSMs in one call --------------------------------------------------------------------------------------
// Several MSMs kept in flight on their own streams time-share the GPU kernel by kernel: a resident k_accumulate owns
// nearly every VGPR (3 waves x 168 registers per SIMD), so the kernels of the other MSMs (prepare, the sorts, the
// tails) wait for wave slots 5-20x longer than they run alone (rocprofv3 trace of 4 in flight at 2^20: k_prepare
// 0.1 -> 2 ms), and every stream's next MSM waits for its own tail.  A batch knows all its MSMs up front:
//   * every prepare + sort goes to two HIGH-PRIORITY streams (memory-bound kernels: they take the wave slots first as
//     accumulation wavefronts retire, and give the VALU back while they wait for memory);
//   * the accumulations (+ combine) run on three normal-priority streams, each waiting only for its own sort -- the
//     next accumulation's wavefronts fill the slots the previous one drains;
//   * ONE tail per round for all jobs (their bucket grids lie end to end): three launches, and no stream waits for it.
// Measured (MI355X, 2^20 points each): 8 / 20 MSMs 1.67 / 1.53 ms per MSM against 1.74 / 1.60 with four single calls
// in flight; level from 40 on (the machine is issue-bound on the total work either way).  Strict
// phase order (all sorts, then all accumulations) measured 6 % slower than this pipeline, tails in groups of 2-10 under
// the later accumulations level, an occupancy cap on k_accumulate (LDS allocation) 3-9 % slower.
// Job j's scratch is a private context (ctx->jobs[j]); results are the bytes of the single-MSM entry point.
// (Measured and removed, round 3: the batch captured and replayed as ONE hipGraph -- 1.2 % slower than this eager
// enqueue, whose host side runs 1.5 ms ahead of a 30 ms batch anyway: profiles/r03_ab_scheduling.txt, git tag exp/many-graph.)
int snarkv::launch_msm_pippenger_many(snarkv_ctx* ctx, size_t count, const void* const* d_s, const void* const* d_p,
                                      const size_t* n, int window_bits, void* d_out, bool partial_out, hipEvent_t* ready) {
  const size_t ostride = partial_out ? SNARKV_G1_PARTIAL_BYTES : 64;
  ctx->last_many_jobs = 0;
  ctx->last_split_workers = 0;
  if (count == 0) return SNARKV_OK;
  uint32_t c0 = 0, w0 = 0, b0 = 0;
  bool uniform = true, large = false;
  size_t nmax = 0, sig = count * 1000003u + (size_t)(uint32_t)window_bits;
  for (size_t i = 0; i < count; ++i) {
    if (n[i] == 0) return SNARKV_ERR_EMPTY;
    uint32_t c, w, b;
    SNARKV_TRY(pip_geometry(n[i], window_bits, &c, &w, &b));
    if (i == 0) c0 = c, w0 = w, b0 = b;
    uniform = uniform && c == c0;
    large = large || pip_chunk_pipeline(n[i], window_bits, false, nullptr);  // the same rule as the single call
    nmax = std::max(nmax, n[i]);
    sig = sig * 31 + n[i];
  }
  const char* em = getenv("SNARKV_MANY_MODE");  // 0: one MSM after the other through the single-call path (A/B knob)
  if (count == 1 || large || (em && atoi(em) == 0) || ctx->is_lane) {
    for (size_t i = 0; i < count; ++i) {
      if (ready) SNARKV_HIP(hipStreamWaitEvent(ctx->stream, ready[i], 0));
      SNARKV_TRY(launch_msm_pippenger_auto(ctx, d_s[i], d_p[i], n[i], window_bits, (uint8_t*)d_out + ostride * i, partial_out));
    }
    return SNARKV_OK;
  }
  // jobs per round: bounded by the scratch footprint (~560 B per point + two bucket grids)
  const size_t per_job = nmax * 600 + (size_t)w0 * b0 * SNARKV_G1_PARTIAL_BYTES * 2 + (1u << 20);
  size_t G = std::min<size_t>(count, SNARKV_MANY_MAX_JOBS);
  G = std::max<size_t>(1, std::min<size_t>(G, ((size_t)48 << 30) / per_job));
  if (const char* eg = getenv("SNARKV_MANY_JOBS")) G = std::max<size_t>(1, std::min<size_t>(G, (size_t)atoi(eg)));
  const size_t rounds = (count + G - 1) / G;
  G = (count + rounds - 1) / rounds;  // even rounds
  SNARKV_TRY(ctx_lanes(ctx));
  while ((size_t)ctx->njobs < G) {
    snarkv_ctx* j = nullptr;
    // a job context is scratch + events only (its phases are enqueued on the scheduler's streams): it borrows this
    // context's stream handle instead of creating a stream of its own -- the runtime maps streams onto its hardware
    // queues in creation order, and dozens of idle streams would shift the mapping of every stream created after them
    SNARKV_TRY(snarkv_ctx_create(ctx->device, (void*)ctx->stream, &j));
    j->is_lane = true;
    j->throughput_mode = true;  // long runs: other accumulations are always resident next to a job's
    SNARKV_HIP(hipEventCreateWithFlags(&j->sorted_ev, hipEventDisableTiming));  // the job's sort is done (its accumulation waits for it)
    j->sorted_ev_ready = true;
    ctx->jobs[ctx->njobs++] = j;
  }
  if (!ctx->hi_ready) {
    int least = 0, greatest = 0;
    SNARKV_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int i = 0; i < 2; ++i) SNARKV_HIP(hipStreamCreateWithPriority(&ctx->hi_stream[i], hipStreamNonBlocking, greatest));
    for (int i = 0; i < 2; ++i) SNARKV_HIP(hipEventCreateWithFlags(&ctx->many_ev[i], hipEventDisableTiming));
    ctx->hi_ready = true;
  }
  const bool tm = ctx->stage_timing;
  if (tm && !ctx->ev_ready) {
    for (int i = 0; i <= SNARKV_PIP_STAGES; ++i) SNARKV_HIP(hipEventCreate(&ctx->ev[i]));
    ctx->ev_ready = true;
  }
  if (sig != ctx->many_sig) {  // a new shape may grow (free + reallocate) scratch that queued work still uses
    SNARKV_HIP(hipDeviceSynchronize());
    ctx->many_sig = sig;
  }
  hipStream_t S[3] = {ctx->stream, ctx->sub[0]->stream, ctx->sub[1]->stream};  // the accumulation streams
  constexpr int nS = 3;  // three of them: 2 / 4 measured level or worse (profiles/r02_sweep_many.txt)
  // the context's stream waits for everything queued on the other four streams (join), then they wait for it (fork)
  auto join_and_fork = [&]() -> int {
    for (int k = 0; k < 2; ++k) {
      SNARKV_HIP(hipEventRecord(ctx->many_ev[k], ctx->hi_stream[k]));
      SNARKV_HIP(hipStreamWaitEvent(ctx->stream, ctx->many_ev[k], 0));
    }
    for (int k = 0; k + 1 < nS; ++k) {
      SNARKV_HIP(hipEventRecord(ctx->sub_ev[k], S[k + 1]));
      SNARKV_HIP(hipStreamWaitEvent(ctx->stream, ctx->sub_ev[k], 0));
    }
    SNARKV_HIP(hipEventRecord(ctx->sub_ev[4], ctx->stream));
    for (int k = 0; k < 2; ++k) SNARKV_HIP(hipStreamWaitEvent(ctx->hi_stream[k], ctx->sub_ev[4], 0));
    for (int k = 0; k + 1 < nS; ++k) SNARKV_HIP(hipStreamWaitEvent(S[k + 1], ctx->sub_ev[4], 0));
    return SNARKV_OK;
  };
  void* d_grids = nullptr;
  const size_t grid_bytes = (size_t)w0 * b0 * SNARKV_G1_PARTIAL_BYTES;
  if (uniform) SNARKV_TRY(ctx_reserve(ctx, SLOT_MGPU_GRID, grid_bytes * G, &d_grids));
  if (tm) SNARKV_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
  SNARKV_TRY(join_and_fork());  // inputs may still be in flight on the caller's stream
  for (size_t lo = 0; lo < count; lo += G) {
    const size_t hi = std::min(count, lo + G);
    const bool last = hi == count;
    for (size_t i = lo; i < hi; ++i) {
      snarkv_ctx* job = ctx->jobs[i - lo];
      job->mont = ctx->mont;
      job->stage_timing = tm && last;
      void* grid = uniform ? (uint8_t*)d_grids + grid_bytes * (i - lo) : nullptr;
      hipStream_t sa = ctx->hi_stream[(i - lo) % 2], sb = S[(i - lo) % nS];
      if (ready) SNARKV_HIP(hipStreamWaitEvent(sa, ready[i], 0));
      SNARKV_TRY(launch_msm_pippenger_phases(job, sa, PIP_PHASE_SORT, d_s[i], d_p[i], n[i], window_bits, nullptr, false,
                                             nullptr, grid));
      SNARKV_HIP(hipEventRecord(job->sorted_ev, sa));
      SNARKV_HIP(hipStreamWaitEvent(sb, job->sorted_ev, 0));
      SNARKV_TRY(launch_msm_pippenger_phases(job, sb, PIP_PHASE_ACC, d_s[i], d_p[i], n[i], window_bits, nullptr, false,
                                             nullptr, grid));
      // a ragged batch (different window sizes) cannot share one tail: each job's own, behind its accumulation
      if (!uniform)
        SNARKV_TRY(launch_msm_pippenger_phases(job, sb, PIP_PHASE_TAIL, d_s[i], d_p[i], n[i], window_bits,
                                               (uint8_t*)d_out + ostride * i, partial_out, nullptr, nullptr));
      job->stage_timing = false;
    }
    SNARKV_TRY(join_and_fork());
    if (uniform) {
      SNARKV_TRY(launch_buckets_reduce_many(ctx, ctx->stream, d_grids, c0, w0, (uint32_t)(hi - lo),
                                            (uint8_t*)d_out + ostride * lo, partial_out));
      if (!last) SNARKV_TRY(join_and_fork());  // the next round overwrites the grids
    }
    if (last) ctx->last_many_jobs = (int)(hi - lo);
  }
  if (tm) SNARKV_HIP(hipEventRecord(ctx->ev[SNARKV_PIP_STAGES - 1], ctx->stream));
  return SNARKV_OK;
}

extern "C" {

int snarkv_g1_msm_pippenger_many_dev(snarkv_ctx* ctx, size_t count, const void* const* d_scalars32,
                                     const void* const* d_points64, const size_t* n, int window_bits, void* d_out64s) {
  if (!ctx || (count && (!d_scalars32 || !d_points64 || !n || !d_out64s))) return SNARKV_ERR_ARG;
  for (size_t i = 0; i < count; ++i)
    if (!d_scalars32[i] || !d_points64[i]) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_msm_pippenger_many(ctx, count, d_scalars32, d_points64, n, window_bits, d_out64s, false);
}

int snarkv_g1_msm_pippenger_many_partial_dev(snarkv_ctx* ctx, size_t count, const void* const* d_scalars32,
                                             const void* const* d_points64, const size_t* n, int window_bits,
                                             void* d_partials) {
  if (!ctx || (count && (!d_scalars32 || !d_points64 || !n || !d_partials))) return SNARKV_ERR_ARG;
  for (size_t i = 0; i < count; ++i)
    if (!d_scalars32[i] || !d_points64[i]) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_msm_pippenger_many(ctx, count, d_scalars32, d_points64, n, window_bits, d_partials, true);
}

// HOST-resident batch: job i's scalars / points are uploaded on a copy stream of their own, one event per job, and the
// batch scheduler makes a job's first kernel wait for ITS event only -- the uploads of jobs i + 1 .. run under the
// kernels of jobs .. i.  A 2^20-point job is 96 MiB: ~1.8 ms over PCIe Gen5 x16 against ~1.5 ms of kernels, so a batch
// runs at the link's rate (bench.py `host_resident`).  Pinned sources (snarkv_ctx_host_buffer, snarkv_host_register) are
// DMA reads; pageable ones go through the runtime's bounce buffer at about a third of the rate.
int snarkv_g1_msm_pippenger_many(snarkv_ctx* ctx, size_t count, const uint8_t* const* scalars32,
                                 const uint8_t* const* points64, const size_t* n, uint32_t flags, uint8_t* out64s) {
  if (!ctx || (count && (!scalars32 || !points64 || !n || !out64s))) return SNARKV_ERR_ARG;
  if (count == 0) return SNARKV_OK;
  size_t total = 0;
  for (size_t i = 0; i < count; ++i) {
    if (!scalars32[i] || !points64[i]) return SNARKV_ERR_ARG;
    if (n[i] == 0) return SNARKV_ERR_EMPTY;
    total += n[i];
  }
  SNARKV_HIP(hipSetDevice(ctx->device));
  SNARKV_CALL_FLAGS(ctx, flags);
  // staging for as many jobs at a time as fit 12 GiB (a 2^20-point job is 96 MiB); more run as successive sub-batches
  const size_t kStageCap = (size_t)12 << 30;
  if (!ctx->copy_ready) {
    SNARKV_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    ctx->copy_ready = true;
  }
  std::vector<hipEvent_t> evs(count, nullptr);
  auto drop_events = [&]() {
    for (auto e : evs)
      if (e) (void)hipEventDestroy(e);
  };
  int rc = SNARKV_OK;
  for (size_t lo = 0; lo < count && rc == SNARKV_OK;) {
    size_t hi = lo, pts = 0;
    while (hi < count && (hi == lo || (pts + n[hi]) * 96 <= kStageCap)) pts += n[hi++];
    void *d_s = nullptr, *d_p = nullptr, *d_out = nullptr;
    rc = ctx_reserve(ctx, SLOT_IN_SCALARS, pts * 32, &d_s);
    if (rc == SNARKV_OK) rc = ctx_reserve(ctx, SLOT_IN_POINTS, pts * 64, &d_p);
    if (rc == SNARKV_OK) rc = ctx_reserve(ctx, SLOT_OUT, (hi - lo) * 64, &d_out);
    if (rc != SNARKV_OK) break;
    std::vector<const void*> ps(hi - lo), pp(hi - lo);
    // the staging buffers may still be read by work queued on the context's stream: the uploads start behind it
    hipEvent_t& fence = evs[lo];  // (re-recorded below as job lo's own event)
    if (!fence && hipEventCreateWithFlags(&fence, hipEventDisableTiming) != hipSuccess) { rc = SNARKV_ERR_DEVICE; break; }
    (void)hipEventRecord(fence, ctx->stream);
    (void)hipStreamWaitEvent(ctx->copy_stream, fence, 0);
    size_t off = 0;
    for (size_t i = lo; i < hi && rc == SNARKV_OK; ++i) {
      ps[i - lo] = (const uint8_t*)d_s + 32 * off;
      pp[i - lo] = (const uint8_t*)d_p + 64 * off;
      if (hipMemcpyAsync((void*)ps[i - lo], scalars32[i], n[i] * 32, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess ||
          hipMemcpyAsync((void*)pp[i - lo], points64[i], n[i] * 64, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess ||
          (!evs[i] && hipEventCreateWithFlags(&evs[i], hipEventDisableTiming) != hipSuccess) ||
          hipEventRecord(evs[i], ctx->copy_stream) != hipSuccess) {
        set_last_error("host-resident batch: upload of job %zu failed: %s", i, hipGetErrorString(hipGetLastError()));
        rc = SNARKV_ERR_DEVICE;
      }
      off += n[i];
    }
    if (rc != SNARKV_OK) break;
    if ((flags | ctx->flags) & SNARKV_FLAG_VALIDATE) {  // validation reads everything: no overlap on this path
      (void)hipStreamSynchronize(ctx->copy_stream);
      rc = check_validate(ctx, d_s, d_p, pts, flags);
      if (rc != SNARKV_OK) break;
    }
    rc = launch_msm_pippenger_many(ctx, hi - lo, ps.data(), pp.data(), n + lo, 0, d_out, false, evs.data() + lo);
    if (rc == SNARKV_OK) rc = fetch_out(ctx, d_out, out64s + 64 * lo, (hi - lo) * 64);
    lo = hi;
  }
  if (rc != SNARKV_OK) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
  }
  drop_events();
  return rc;
}

// hipHostRegister / hipHostUnregister for callers without a HIP binding: pins `bytes` at `p` (e.g. a Vec<G1Affine>'s
// buffer) so that the uploads above are DMA reads straight out of it
int snarkv_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return SNARKV_OK;
}
int snarkv_host_unregister(void* p) {
  if (!p) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipHostUnregister(p));
  return SNARKV_OK;
}

int snarkv_g1_msm_pippenger(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                            uint32_t flags, uint8_t out64[64]) {
  if (!ctx || !scalars32 || !points64 || !out64) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_ERR_EMPTY;  // reference panics: msm.rs:265
  SNARKV_HIP(hipSetDevice(ctx->device));
  SNARKV_CALL_FLAGS(ctx, flags);
  void *d_s, *d_p, *d_out;
  SNARKV_TRY(stage_in(ctx, SLOT_IN_SCALARS, scalars32, n * 32, &d_s));
  SNARKV_TRY(stage_in(ctx, SLOT_IN_POINTS, points64, n * 64, &d_p));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, 64, &d_out));
  SNARKV_TRY(check_validate(ctx, d_s, d_p, n, flags));
  SNARKV_TRY(pippenger_maybe_split(ctx, d_s, d_p, n, 0, d_out, false));
  return fetch_out(ctx, d_out, out64, 64);
}

int snarkv_g1_msm_pippenger_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64, size_t n,
                                int window_bits, void* d_out64) {
  if (!ctx || !d_scalars32 || !d_points64 || !d_out64) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_ERR_EMPTY;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return pippenger_maybe_split(ctx, d_scalars32, d_points64, n, window_bits, d_out64, false);
}

int snarkv_g1_msm_pippenger_partial_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64,
                                        size_t n, int window_bits, void* d_partial) {
  if (!ctx || !d_scalars32 || !d_points64 || !d_partial) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_ERR_EMPTY;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return pippenger_maybe_split(ctx, d_scalars32, d_points64, n, window_bits, d_partial, true);
}

int snarkv_g1_fold_partials_dev(snarkv_ctx* ctx, const void* d_partials, size_t count, void* d_out64) {
  if (!ctx || !d_partials || !d_out64) return SNARKV_ERR_ARG;
  if (count == 0) return SNARKV_ERR_EMPTY;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_fold_partials(ctx, d_partials, count, d_out64);
}

int snarkv_g1_fold_partials_many_dev(snarkv_ctx* ctx, const void* d_partials, size_t count, size_t jobs, void* d_out64s) {
  if (!ctx || !d_partials || !d_out64s) return SNARKV_ERR_ARG;
  if (count == 0 || jobs == 0) return SNARKV_ERR_EMPTY;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_fold_partials_many(ctx, d_partials, count, jobs, d_out64s);
}

int snarkv_g1_msm_bucket_geometry(size_t n_total, int window_bits, uint32_t* c, uint32_t* windows,
                                  uint32_t* buckets_per_window) {
  if (!c || !windows || !buckets_per_window) return SNARKV_ERR_ARG;
  if (n_total == 0) return SNARKV_ERR_EMPTY;
  return pip_geometry(n_total, window_bits, c, windows, buckets_per_window);
}

int snarkv_g1_msm_fill_buckets_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64, size_t n,
                                   int window_bits, void* d_buckets) {
  if (!ctx || !d_scalars32 || !d_points64 || !d_buckets || window_bits < 2 || window_bits > 22) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_ERR_EMPTY;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_msm_pippenger(ctx, d_scalars32, d_points64, n, window_bits, nullptr, false, d_buckets);
}

int snarkv_g1_buckets_add_dev(snarkv_ctx* ctx, void* d_dst, const void* d_src, size_t count) {
  if (!ctx || !d_dst || !d_src) return SNARKV_ERR_ARG;
  if (count == 0) return SNARKV_ERR_EMPTY;
  if (count >= ((size_t)1 << 31)) return SNARKV_ERR_LENGTH;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_buckets_add(ctx, d_dst, d_src, count);
}

int snarkv_g1_buckets_reduce_dev(snarkv_ctx* ctx, const void* d_buckets, uint32_t c, uint32_t w0, uint32_t wcount,
                                 void* d_partial) {
  if (!ctx || !d_buckets || !d_partial || c < 2 || c > 22) return SNARKV_ERR_ARG;
  if (wcount == 0) return SNARKV_ERR_EMPTY;
  uint32_t cc, windows, bpw;
  SNARKV_TRY(pip_geometry(1, (int)c, &cc, &windows, &bpw));
  if ((uint64_t)w0 + wcount > windows) return SNARKV_ERR_LENGTH;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_buckets_reduce(ctx, d_buckets, c, w0, wcount, d_partial);
}

int snarkv_dk_create(snarkv_ctx* ctx, const uint8_t g1_64[64], const uint8_t g2_128[128],
                     const uint8_t s_g2_128[128], uint32_t flags, snarkv_dk** out) {
  if (!ctx || !g1_64 || !g2_128 || !s_g2_128 || !out) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipSetDevice(ctx->device));
  SNARKV_CALL_FLAGS(ctx, flags);
  flags |= ctx->flags;
  uint8_t both[256];
  memcpy(both, g2_128, 128);
  memcpy(both + 128, s_g2_128, 128);
  void* d_in;
  SNARKV_TRY(stage_in(ctx, SLOT_IN_POINTS, both, 256, &d_in));
  // staging source is a stack buffer: finish the copy before returning
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  if (flags & SNARKV_FLAG_VALIDATE) {
    int bad = 0;
    SNARKV_TRY(launch_validate_g2(ctx, d_in, &bad));
    if (bad) {
      set_last_error("deciding key: G2 point non-canonical or off the twist");
      return SNARKV_ERR_ENCODING;
    }
    void* d_g1;
    SNARKV_TRY(stage_in(ctx, SLOT_IN_SCALARS, g1_64, 64, &d_g1));
    SNARKV_TRY(launch_validate(ctx, nullptr, d_g1, 1, &bad));
    if (bad) {
      set_last_error("deciding key: G1 generator non-canonical or off-curve");
      return SNARKV_ERR_ENCODING;
    }
  }
  snarkv_dk* dk = new snarkv_dk();
  dk->device = ctx->device;
  memcpy(dk->g1, g1_64, 64);
  hipError_t e = hipMalloc(&dk->d_prep, 2 * g2_prepared_bytes());
  if (e != hipSuccess) {
    set_last_error("hipMalloc(dk): %s", hipGetErrorString(e));
    delete dk;
    return SNARKV_ERR_DEVICE;
  }
  int rc = launch_g2_prepare(ctx, d_in, dk->d_prep);
  if (rc == SNARKV_OK) {
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e2 != hipSuccess) {
      set_last_error("g2_prepare: %s", hipGetErrorString(e2));
      rc = SNARKV_ERR_DEVICE;
    }
  }
  if (rc < 0) {
    (void)hipFree(dk->d_prep);
    delete dk;
    return rc;
  }
  *out = dk;
  return SNARKV_OK;
}

void snarkv_dk_destroy(snarkv_dk* dk) {
  if (!dk) return;
  (void)hipSetDevice(dk->device);
  if (dk->d_prep) (void)hipFree(dk->d_prep);
  delete dk;
}

int snarkv_kzg_decide_batch_dev(snarkv_ctx* ctx, const snarkv_dk* dk, const void* d_accs128, size_t m,
                                void* d_ok) {
  if (!ctx || !dk || !d_accs128 || !d_ok) return SNARKV_ERR_ARG;
  if (m == 0) return SNARKV_ERR_EMPTY;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_decide(ctx, dk->d_prep, d_accs128, m, d_ok, nullptr);
}

int snarkv_kzg_decide_batch(snarkv_ctx* ctx, const snarkv_dk* dk, const uint8_t* accs128, size_t m,
                            uint32_t flags, uint8_t* ok) {
  if (!ctx || !dk || !accs128 || !ok) return SNARKV_ERR_ARG;
  if (m == 0) return 1;  // decide_all over an empty list is Ok(()) (decider.rs:84-93)
  SNARKV_HIP(hipSetDevice(ctx->device));
  SNARKV_CALL_FLAGS(ctx, flags);
  flags |= ctx->flags;
  void *d_a, *d_ok;
  SNARKV_TRY(stage_in(ctx, SLOT_IN_POINTS, accs128, m * 128, &d_a));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, m, &d_ok));
  if (flags & SNARKV_FLAG_VALIDATE) {
    int bad = 0;
    SNARKV_TRY(launch_validate(ctx, nullptr, d_a, 2 * m, &bad));
    if (bad) {
      set_last_error("%d accumulator points non-canonical or off-curve", bad);
      return SNARKV_ERR_ENCODING;
    }
  }
  SNARKV_TRY(launch_decide(ctx, dk->d_prep, d_a, m, d_ok, nullptr));
  SNARKV_TRY(fetch_out(ctx, d_ok, ok, m));
  int all = 1;
  for (size_t i = 0; i < m; ++i) all &= ok[i] ? 1 : 0;
  return all;
}

int snarkv_kzg_decide(snarkv_ctx* ctx, const snarkv_dk* dk, const uint8_t acc128[128], uint32_t flags) {
  uint8_t ok = 0;
  int rc = snarkv_kzg_decide_batch(ctx, dk, acc128, 1, flags, &ok);
  if (rc < 0) return rc;
  return ok ? 1 : 0;
}

int snarkv_kzg_pairing_value(snarkv_ctx* ctx, const snarkv_dk* dk, const uint8_t acc128[128], uint8_t gt384[384]) {
  if (!ctx || !dk || !acc128 || !gt384) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipSetDevice(ctx->device));
  void *d_a, *d_gt;
  SNARKV_TRY(stage_in(ctx, SLOT_IN_POINTS, acc128, 128, &d_a));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, 384, &d_gt));
  SNARKV_TRY(launch_decide(ctx, dk->d_prep, d_a, 1, nullptr, d_gt));
  return fetch_out(ctx, d_gt, gt384, 384);
}

int snarkv_sample_scalars_dev(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_scalars32) {
  if (!ctx || !d_scalars32) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_OK;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_sample_scalars(ctx, seed, first, n, d_scalars32);
}

int snarkv_sample_points_dev(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_points64) {
  if (!ctx || !d_points64) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_OK;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_sample_points(ctx, seed, first, n, d_points64);
}

int snarkv_ubench_valu(snarkv_ctx* ctx, int which, int iters, double* ops_per_s) {
  if (!ctx || !ops_per_s || iters <= 0 || which < 0 || which > 1) return SNARKV_ERR_ARG;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_ubench(ctx, which, iters, ops_per_s);
}

int snarkv_g1_validate(snarkv_ctx* ctx, const uint8_t* points64, size_t n) {
  if (!ctx || !points64) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_OK;
  SNARKV_HIP(hipSetDevice(ctx->device));
  void* d_p;
  SNARKV_TRY(stage_in(ctx, SLOT_IN_POINTS, points64, n * 64, &d_p));
  int bad = 0;
  SNARKV_TRY(launch_validate(ctx, nullptr, d_p, n, &bad));
  if (bad) {
    set_last_error("%d of %zu points are non-canonical or off-curve", bad, n);
    return SNARKV_ERR_ENCODING;
  }
  return SNARKV_OK;
}

// ---- context-free entry points -------------------------------------------
int bn254_set_flags(uint32_t flags) {
  if (flags & ~(SNARKV_FLAG_VALIDATE | SNARKV_FLAG_MONTGOMERY)) return SNARKV_ERR_ARG;
  g_pool.flags.store(flags);  // applied to a pool context when a call checks it out; calls in flight keep theirs
  return SNARKV_OK;
}

uint32_t bn254_get_flags(void) { return tl_flags >= 0 ? (uint32_t)tl_flags : g_pool.flags.load(); }

int64_t bn254_set_thread_flags(int64_t flags) {
  const int64_t before = tl_flags;
  // unknown bits, or a negative value other than -1 (e.g. an error code of this very function handed back as "the
  // previous value"): refused, the override stays what it was
  if (flags < -1 || (flags >= 0 && (flags & ~(int64_t)(SNARKV_FLAG_VALIDATE | SNARKV_FLAG_MONTGOMERY)))) return SNARKV_ERR_ARG;
  tl_flags = flags;
  return before;
}

int bn254_default_contexts(int* created, int* cap) {
  std::lock_guard<std::mutex> lk(g_pool.mu);
  if (g_pool.cap == 0) g_pool.cap = pool_cap();
  if (created) *created = (int)g_pool.ctx.size();
  if (cap) *cap = g_pool.cap;
  return SNARKV_OK;
}

int bn254_g1_validate(const uint8_t* points64, size_t n) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_g1_validate(c, points64, n);
}

int bn254_kzg_dk_create(const uint8_t g1_64[64], const uint8_t g2_128[128], const uint8_t s_g2_128[128],
                        snarkv_dk** out) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_dk_create(c, g1_64, g2_128, s_g2_128, 0, out);
}

int bn254_kzg_dk_decide_batch(const snarkv_dk* dk, const uint8_t* accs128, size_t m, uint8_t* ok) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_kzg_decide_batch(c, dk, accs128, m, 0, ok);
}

int bn254_g1_msm_naive(const uint8_t* scalars32, const uint8_t* points64, size_t n, uint8_t out64[64]) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_g1_msm_naive(c, scalars32, points64, n, 0, out64);
}

int bn254_g1_msm_batched(const uint8_t* scalars32, const uint8_t* points64, const uint32_t* offsets, size_t n_msm,
                         uint8_t* out) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_g1_msm_batched(c, scalars32, points64, offsets, n_msm, 0, out);
}

int bn254_g1_decompress(const uint8_t* in32, size_t n, uint8_t* out64, uint8_t* ok) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_g1_decompress(c, in32, n, out64, ok);
}

int bn254_host_buffer(int slot, size_t bytes, void** out) {
  if (slot < 0 || slot >= SNARKV_HOST_BUFFERS || !out) return SNARKV_ERR_ARG;
  if (!tl_hb.p) {
    std::lock_guard<std::mutex> lk(g_hb_mu);
    if (!g_hb_free.empty()) {
      tl_hb.p = g_hb_free.back();
      g_hb_free.pop_back();
    } else {
      tl_hb.p = new ThreadHostBufs();
    }
  }
  ThreadHostBufs* hb = tl_hb.p;
  if (bytes == 0) bytes = 16;
  if (hb->cap[slot] < bytes) {
    {
      SNARKV_DEFAULT_LEASE(c);  // (a HIP device must exist; the allocation itself is not tied to a context)
      SNARKV_HIP(hipSetDevice(c->device));
    }
    const size_t old_cap = hb->cap[slot];
    if (hb->buf[slot]) SNARKV_HIP(hipHostFree(hb->buf[slot]));
    hb->buf[slot] = nullptr;
    hb->cap[slot] = 0;
    // (no copy out of the old buffer can be queued: the context-free calls return after their result is on the host)
    // geometric growth: hipHostFree / hipHostMalloc synchronise the whole device (other pool contexts are in flight), so a
    // thread whose requests creep up must not reallocate on every call
    const size_t cap = std::max(bytes + bytes / 4 + 4096, 2 * old_cap);
    SNARKV_HIP(hipHostMalloc(&hb->buf[slot], cap, hipHostMallocDefault));
    hb->cap[slot] = cap;
  }
  *out = hb->buf[slot];
  return SNARKV_OK;
}

// Everything the context-free entry points keep for the life of the process, released: the pool's contexts (streams +
// scratch), the deciding-key line tables, the pinned buffer sets of the calling thread and of threads that have ended.
// Refused (SNARKV_ERR_ARG) while any context-free call is in flight.  The next bn254_* call starts over lazily.
int bn254_shutdown(void) {
  {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    if (tl_depth > 0) {
      set_last_error("bn254_shutdown: called from inside a context-free call");
      return SNARKV_ERR_ARG;
    }
    for (char b : g_pool.busy)
      if (b) {
        set_last_error("bn254_shutdown: a context-free call is in flight on another thread");
        return SNARKV_ERR_ARG;
      }
    for (snarkv_ctx* c : g_pool.ctx) snarkv_ctx_destroy(c);
    g_pool.ctx.clear();
    g_pool.busy.clear();
  }
  tl_slot = -1;
  {
    std::lock_guard<std::mutex> lk(g_dk_cache_mu);
    for (auto& e : g_dk_cache) e.dk.reset();
  }
  std::vector<ThreadHostBufs*> sets;
  {
    std::lock_guard<std::mutex> lk(g_hb_mu);
    sets.swap(g_hb_free);
  }
  if (tl_hb.p) sets.push_back(tl_hb.p), tl_hb.p = nullptr;
  for (ThreadHostBufs* hb : sets) {
    for (int i = 0; i < SNARKV_HOST_BUFFERS; ++i)
      if (hb->buf[i]) (void)hipHostFree(hb->buf[i]);
    delete hb;
  }
  (void)hipGetLastError();
  return SNARKV_OK;
}

int bn254_g1_msm_pippenger(const uint8_t* scalars32, const uint8_t* points64, size_t n, uint8_t out64[64]) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_g1_msm_pippenger(c, scalars32, points64, n, 0, out64);
}

int bn254_kzg_decide_batch(const uint8_t g1_64[64], const uint8_t g2_128[128], const uint8_t s_g2_128[128],
                           const uint8_t* accs128, size_t m, uint8_t* ok) {
  SNARKV_DEFAULT_LEASE(c);
  if (!g1_64 || !g2_128 || !s_g2_128) return SNARKV_ERR_ARG;
  // The reference rebuilds `G2Prepared` on every decide (decider.rs:74); a verifier decides against ONE key (or a few:
  // one per recursion layer / encoding), so the line tables of the last DK_CACHE keys are kept; the 320 key bytes + the
  // encoding they were given in are the tag.  Shared by the pool's contexts (the tables are read-only device memory).
  // The lock covers the lookup and the publication only: a miss builds its tables (G2 preparation on the GPU + a
  // synchronisation) OUTSIDE it, so concurrent callers with other keys -- or the same key: the loser of the race drops
  // its copy -- never queue behind a build.  A replaced key's tables are freed when the last call using
  // them returns (shared_ptr).
  uint8_t now[321];
  memcpy(now, g1_64, 64), memcpy(now + 64, g2_128, 128), memcpy(now + 192, s_g2_128, 128);
  now[320] = c->mont ? 1 : 0;
  auto& cache_mu = g_dk_cache_mu;
  auto& cache = g_dk_cache;
  auto& tick = g_dk_tick;
  auto lookup = [&]() -> std::shared_ptr<snarkv_dk> {  // (under cache_mu)
    for (auto& e : cache)
      if (e.dk && memcmp(e.tag, now, sizeof now) == 0) {
        e.used = ++tick;
        return e.dk;
      }
    return nullptr;
  };
  std::shared_ptr<snarkv_dk> dk;
  {
    std::lock_guard<std::mutex> lk(cache_mu);
    dk = lookup();
  }
  if (!dk) {
    snarkv_dk* fresh = nullptr;
    SNARKV_TRY(snarkv_dk_create(c, g1_64, g2_128, s_g2_128, 0, &fresh));
    std::shared_ptr<snarkv_dk> mine(fresh, [](snarkv_dk* p) { snarkv_dk_destroy(p); });
    std::lock_guard<std::mutex> lk(cache_mu);
    dk = lookup();  // somebody else published the same key meanwhile: use theirs, `mine` is freed on return
    if (!dk) {
      DkCacheEntry* victim = &cache[0];
      for (auto& e : cache) {  // a free slot, else the least recently used
        if (!e.dk) {
          victim = &e;
          break;
        }
        if (e.used < victim->used) victim = &e;
      }
      victim->dk = mine;
      memcpy(victim->tag, now, sizeof now);
      victim->used = ++tick;
      dk = mine;
    }
  }
  return snarkv_kzg_decide_batch(c, dk.get(), accs128, m, 0, ok);
}

int bn254_kzg_decide(const uint8_t g1_64[64], const uint8_t g2_128[128], const uint8_t s_g2_128[128],
                     const uint8_t acc128[128]) {
  uint8_t ok = 0;
  int rc = bn254_kzg_decide_batch(g1_64, g2_128, s_g2_128, acc128, 1, &ok);
  if (rc < 0) return rc;
  return ok ? 1 : 0;
}

int bn254_poseidon_create(uint32_t t, uint32_t rate, uint32_t r_f, uint32_t r_p, const uint8_t* start,
                          const uint8_t* partial, const uint8_t* end, const uint8_t* mds, const uint8_t* pre_sparse_mds,
                          const uint8_t* sparse_rows, const uint8_t* sparse_col_hats, snarkv_poseidon** out) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_poseidon_create(c, t, rate, r_f, r_p, start, partial, end, mds, pre_sparse_mds, sparse_rows,
                                sparse_col_hats, out);
}

int bn254_ipa_dk_create(const uint8_t* g_points64, size_t n, snarkv_ipa_dk** out) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_ipa_dk_create(c, g_points64, n, out);
}

int bn254_ipa_decide_batch(const snarkv_ipa_dk* dk, const uint8_t* xi32, const uint8_t* u64, size_t m, uint8_t* ok) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_ipa_decide_batch(c, dk, xi32, u64, m, ok);
}

int bn254_poseidon_transcript_batch(const snarkv_poseidon* ps, const uint8_t* elems, size_t n, size_t L,
                                    const uint32_t* seg_len, size_t S, uint8_t* out) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_poseidon_transcript_batch(c, ps, elems, n, L, seg_len, S, out);
}

int bn254_poseidon_read_batch(const snarkv_poseidon* ps, const uint8_t* proofs, size_t n, size_t stride, const uint8_t* lead,
                              size_t n_lead, const uint32_t* layout, size_t L, const uint32_t* point_offsets, size_t P,
                              const uint32_t* seg_len, size_t S, uint8_t* challenges, uint8_t* points64, uint8_t* ok) {
  SNARKV_DEFAULT_LEASE(c);
  return snarkv_poseidon_read_batch(c, ps, proofs, n, stride, lead, n_lead, layout, L, point_offsets, P, seg_len, S, challenges,
                                    points64, ok);
}

}  // extern "C"
