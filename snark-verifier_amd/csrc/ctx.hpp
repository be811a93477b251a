// Internal: context object behind the C ABI (include/snarkv_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/snarkv_amd.h"
#include "curve_consts.h"

// extern "C" names of the units shared between the BN254 library and the pasta build
#if defined(SNARKV_CURVE_PALLAS)
#define SNARKV_API(name) snarkv_pallas_##name
#else
#define SNARKV_API(name) snarkv_##name
#endif

namespace snarkv {

void set_last_error(const char* fmt, ...);

#define SNARKV_HIP(expr)                                                                   \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::snarkv::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return SNARKV_ERR_DEVICE;                                                            \
    }                                                                                      \
  } while (0)

#define SNARKV_TRY(expr)      \
  do {                        \
    int _rc = (expr);         \
    if (_rc < 0) return _rc;  \
  } while (0)

// Scratch slots (grow-only device buffers owned by the context).
enum Slot {
  SLOT_IN_SCALARS = 0,
  SLOT_IN_POINTS,
  SLOT_IN_OFFSETS,
  SLOT_OUT,
  SLOT_POINTS_MONT,
  SLOT_COUNTS,
  SLOT_OFFSETS,
  SLOT_CURSOR,
  SLOT_BLOCKSUMS,
  SLOT_ENTRIES,
  SLOT_SEG_IDS,
  SLOT_SEG_PARTIALS,
  SLOT_BUCKETS,
  SLOT_CHUNK_PARTIALS,
  SLOT_WINDOW_SUMS,
  SLOT_TERM_PARTIALS,
  SLOT_FLAGS,
  SLOT_MISC,
  SLOT_SORT_TMP,
  SLOT_SHIFTED,
  SLOT_GLV,
  SLOT_BIG_LIST,
  SLOT_MISC2,
  SLOT_TERM_CHAIN,
  SLOT_TERM_MAGS,
  SLOT_SPLIT_PARTIALS,
  SLOT_IPA_XI,
  SLOT_IPA_H,
  SLOT_IPA_OUT,
  SLOT_MGPU_GRID,
  SLOT_MGPU_RECV,
  SLOT_COUNT
};

}  // namespace snarkv

struct snarkv_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  void* buf[snarkv::SLOT_COUNT];
  size_t cap[snarkv::SLOT_COUNT];
  // pinned host buffers handed to the caller (snarkv_ctx_host_buffer): inputs packed there reach the device by DMA
  // instead of the runtime's bounce copy of pageable memory
  void* hbuf[SNARKV_HOST_BUFFERS];
  size_t hbuf_cap[SNARKV_HOST_BUFFERS];
  uint32_t flags;  // default flags of the context (snarkv_ctx_set_flags), OR-ed into every call's own
  bool mont;       // SNARKV_FLAG_MONTGOMERY in effect for the call being enqueued (set by the entry points)
  bool stage_timing;
  float stage_ms[SNARKV_PIP_STAGES];
  hipEvent_t ev[SNARKV_PIP_STAGES + 1];
  bool ev_ready;
  // large MSMs run as pipelined 2^20-point chunks on private sub-contexts (capi.hip)
  snarkv_ctx* sub[4];
  hipEvent_t sub_ev[5];
  bool sub_ready;
  hipEvent_t sorted_ev;  // job contexts of a batch: this job's prepare + sort is done (its accumulation waits for it)
  bool sorted_ev_ready;
  int last_split_workers;  // > 0: the last Pippenger ran as a chunk pipeline on that many worker lanes (stage timing)
  bool throughput_mode;  // several MSMs are kept in flight next to this context's (snarkv_ctx_set_throughput_hint; always on lanes)
  int throughput_peers;  // how many launches share the GPU under the hint (0 = unknown: 16, what bench.py keeps in flight)
  bool is_lane;          // a private sub-context of another context (never starts lanes of its own)
  // job contexts of the batch entry point (snarkv_g1_msm_pippenger_many_dev): scratch of one MSM each
  snarkv_ctx* jobs[SNARKV_MANY_MAX_JOBS];
  int njobs;
  int last_many_jobs;    // > 0: the last call was a batch of that many jobs per round (stage timing reads the jobs' events)
  size_t many_sig;       // shape of the last batch (a new shape may grow scratch: synchronise first)
  hipStream_t copy_stream;   // host-resident batches (snarkv_g1_msm_pippenger_many): the uploads, one event per job
  bool copy_ready;
  hipEvent_t many_ev[2];
  hipEvent_t order_ev;  // snarkv_ctx_wait_stream / snarkv_stream_wait_ctx: the record-and-wait event
  bool order_ev_ready;
  hipStream_t hi_stream[2];  // high-priority streams of the batch pipeline (the sorts) + their join events (many_ev)
  bool hi_ready;
};

struct snarkv_dk {
  int device;
  void* d_prep;  // 2 x G2Prepared29 (g2, -s_g2): the line tables the decide kernels read
  uint8_t g1[64];
};

namespace snarkv {

// The encoding of a call = the context's default flags | the call's own: kept in ctx->mont while the call enqueues its
// kernels (every launcher reads it), restored on the way out.
struct CallFlags {
  snarkv_ctx* c;
  bool saved;
  CallFlags(snarkv_ctx* ctx, uint32_t call_flags) : c(ctx), saved(ctx ? ctx->mont : false) {
    if (c) c->mont = ((c->flags | call_flags) & SNARKV_FLAG_MONTGOMERY) != 0;
  }
  ~CallFlags() {
    if (c) c->mont = saved;
  }
};
#define SNARKV_CALL_FLAGS(ctx, f) ::snarkv::CallFlags _call_flags((ctx), (f))
// Entry points SNARKV_FLAG_MONTGOMERY does NOT cover (the IPA, the Poseidon transcripts: include/snarkv_amd.h): their
// kernels run in the wire form whatever the context's default says
struct WireFormScope {
  snarkv_ctx* c;
  bool saved;
  explicit WireFormScope(snarkv_ctx* ctx) : c(ctx), saved(ctx ? ctx->mont : false) {
    if (c) c->mont = false;
  }
  ~WireFormScope() {
    if (c) c->mont = saved;
  }
};
#define SNARKV_WIRE_FORM(ctx) ::snarkv::WireFormScope _wire_form((ctx))

// Ensure slot capacity; returns device pointer through *out.
int ctx_reserve(snarkv_ctx* ctx, int slot, size_t bytes, void** out);
// four lanes (the context's stream + three private sub-contexts) for independent launches: ctx_impl.inc
int ctx_lanes(snarkv_ctx* ctx);
int ctx_lanes_fork(snarkv_ctx* ctx);
int ctx_lanes_join(snarkv_ctx* ctx);
inline snarkv_ctx* ctx_lane(snarkv_ctx* ctx, size_t i) { return (i % 4 == 3) ? ctx : ctx->sub[i % 4]; }

// kernels' host-side launchers (each enqueues on ctx->stream)
int launch_msm_batched(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, const void* d_offsets,
                       size_t n_msm, size_t n_terms, void* d_out);
int launch_g1_decompress(snarkv_ctx* ctx, const void* d_in32, size_t n, void* d_out64, void* d_ok);
int launch_msm_pippenger(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int window_bits,
                         void* d_out, bool partial_out, void* d_buckets_out = nullptr);
// one Pippenger cut into phases, each on a stream of the caller's choice (msm_pippenger.hip); `ctx` owns the scratch
enum { PIP_PHASE_SORT = 1, PIP_PHASE_ACC = 2, PIP_PHASE_TAIL = 4, PIP_PHASE_ALL = 7 };
int launch_msm_pippenger_phases(snarkv_ctx* ctx, hipStream_t st, int phases, const void* d_scalars, const void* d_points,
                                size_t n, int window_bits, void* d_out, bool partial_out, void* d_buckets_out,
                                void* d_grid);
int launch_buckets_reduce_many(snarkv_ctx* ctx, hipStream_t st, const void* d_grids, uint32_t c, uint32_t windows,
                               uint32_t jobs, void* d_out, bool partial_out);
// `count` independent MSMs, phase-ordered over private job contexts (capi.hip)
// `ready` (optional): one event per job -- its inputs are in place (uploads of a host-resident batch); a job's first kernel
// waits for its event only, so the uploads of later jobs run under the kernels of earlier ones
int launch_msm_pippenger_many(snarkv_ctx* ctx, size_t count, const void* const* d_scalars, const void* const* d_points,
                              const size_t* n, int window_bits, void* d_out, bool partial_out, hipEvent_t* ready = nullptr);
// the product path of a large MSM: single launch, or the chunk pipeline over shared bucket grids (capi.hip)
int launch_msm_pippenger_auto(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int window_bits,
                              void* d_out, bool partial_out);
// does an n-point MSM run as the chunk pipeline over shared bucket grids? (the one rule: capi.hip)
bool pip_chunk_pipeline(size_t n, int window_bits, bool is_lane, size_t* chunk);
int pip_geometry(size_t n_total, int window_bits, uint32_t* c, uint32_t* windows, uint32_t* buckets_per_window);
int launch_buckets_add(snarkv_ctx* ctx, void* d_dst, const void* d_src, size_t count);
int launch_buckets_reduce(snarkv_ctx* ctx, const void* d_buckets, uint32_t c, uint32_t w0, uint32_t wcount, void* d_partial);
int launch_fold_partials(snarkv_ctx* ctx, const void* d_partials, size_t count, void* d_out64, bool partial_out = false);
int launch_fold_partials_many(snarkv_ctx* ctx, const void* d_partials, size_t count, size_t jobs, void* d_out64s);
int launch_validate(snarkv_ctx* ctx, const void* d_scalars, const void* d_points, size_t n, int* bad_host);
int launch_g2_prepare(snarkv_ctx* ctx, const void* d_g2x2_256, void* d_prep);
int launch_decide(snarkv_ctx* ctx, const void* d_prep, const void* d_accs, size_t m, void* d_ok, void* d_gt);
int launch_validate_g2(snarkv_ctx* ctx, const void* d_g2x2_256, int* bad_host);
size_t g2_prepared_bytes();
int launch_sample_scalars(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_out);
int launch_sample_points(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_out);
int launch_ubench(snarkv_ctx* ctx, int which, int iters, double* ops_per_s);

}  // namespace snarkv
