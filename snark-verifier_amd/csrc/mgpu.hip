// Single-process multi-GPU entry points of the C ABI (include/snarkv_amd.h, "multi-GPU"): what SURVEY.md 8(b) lists as
// `*_multi_gpu` -- for a caller that is ONE process without torchrun (the reference's `NativeLoader` is a unit struct with
// static dispatch, loader.rs:108 / native.rs:11-19: a Rust or C caller cannot be handed a torch.distributed world).
//
// A `snarkv_mgpu` owns one context + HIP stream per listed device.  The same device may be listed several times: each
// entry is a "rank" with its own context, which is how the 1-GPU test box exercises every code path of an 8-rank run.
//
//   snarkv_g1_msm_pippenger_mgpu[_dev]   the reference's own chunking (`chunk = ceil(n / threads)`, every chunk a full
//                                        Pippenger, results added: util/msm.rs:311-336) with GPUs in place of rayon
//                                        threads.  variant 0 = point-sharded (default): rank g reduces its shard to a
//                                        144-byte projective partial; the partials travel to rank 0's device by
//                                        hipMemcpyPeerAsync (xGMI point-to-point; 144 B per rank: latency-bound) and are
//                                        folded there.  variant 1 = bucket-sharded ("bucket-sum allreduce", SURVEY 8e /
//                                        BASELINE config 4): every rank fills the GLOBAL bucket grid from its points, the
//                                        grid is exchanged by window range (peer copies: a reduce-scatter whose reduction
//                                        is EC addition, done on the receiving device), every rank reduces the windows it
//                                        owns, partials gathered and folded as above.
//   snarkv_kzg_decide_batch_mgpu         `decide_all` (pcs/kzg/decider.rs:84-93) with the accumulators sharded over the
//                                        ranks; no exchange (independent pairings), m verdict bytes gathered by the host.
//
// Transports of the 144-byte partials (snarkv_mgpu_set_transport):
//   SNARKV_MGPU_TRANSPORT_PEER_COPY (default)  hipMemcpyPeerAsync to rank 0's device, fold there, the 64-byte result
//                                              broadcast back so that EVERY rank's device holds it (all-reduce semantics,
//                                              SURVEY.md 8e); no collective library involved
//   SNARKV_MGPU_TRANSPORT_RCCL                 `ncclCommInitAll` over the handle's devices + ONE grouped `ncclAllGather`
//                                              of 144 B per rank over xGMI -- what BASELINE's north_star names -- after
//                                              which every rank folds all partials itself.  librccl is resolved at run
//                                              time (dlopen: the copy already in the process, e.g. torch's, else
//                                              /opt/rocm/lib) so the library carries no link-time dependency on it; RCCL
//                                              refuses a device listed twice, so emulated ranks cannot use it.
// (One process per GPU keeps using RCCL through torch.distributed: snark-verifier_amd/distributed.py.)
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "ctx.hpp"

struct snarkv_mgpu {
  std::vector<snarkv_ctx*> ctx;     // one per rank
  std::vector<hipEvent_t> done;     // rank's work so far finished (recorded on its stream when a peer needs it)
  std::vector<hipEvent_t> filled;   // bucket-sharded: rank's grid is complete
  std::vector<void*> d_gather;      // on rank 0's device: world x 144 B
  std::vector<void*> d_part;        // per rank: its 144-byte partial
  std::vector<void*> d_result;      // per rank: the 64-byte affine result of the last MSM (all-reduce semantics)
  std::vector<void*> d_gather_all;  // RCCL transport: per rank world x 144 B
  // the batch entry point (snarkv_g1_msm_pippenger_many_mgpu_dev): per rank `many_cap` jobs' worth of
  std::vector<void*> d_parts_many;   //   its own partials                 [job][144]
  std::vector<void*> d_gather_many;  //   every rank's partials as gathered [rank][job][144]   (rank 0 only with peer copies)
  std::vector<void*> d_byjob_many;   //   ... transposed for the fold       [job][rank][144]   (rank 0 only with peer copies)
  std::vector<void*> d_results_many; //   the affine results                [job][64]
  size_t many_cap = 0;
  std::vector<snarkv_dk*> dk;       // decide: one prepared key per rank (lazily, keyed by the last key bytes)
  uint8_t dk_bytes[320];
  bool dk_valid = false;
  int transport = 0;
  std::vector<void*> comms;         // RCCL transport: one ncclComm_t per rank (ncclCommInitAll)
  int peer_enabled = 0, peer_unavailable = 0, peer_failed = 0;  // directed pairs of DISTINCT devices
};

namespace {
// the five RCCL entry points of the transport, resolved at run time; rccl.h is not needed for these signatures
struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static std::string g_rccl_why;  // why the library could not be loaded (dlerror() clears itself: read once, here)
RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names)
      if ((api.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;  // a copy already in the process (torch's) first
    for (const char* nm : names) {
      if (api.lib) break;
      api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (!api.lib) {
        const char* e = dlerror();
        if (e) g_rccl_why = e;
      }
    }
    if (api.lib) {
      api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
      api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
      api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
      api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
      api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    }
  });
  return (api.lib && api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd) ? &api : nullptr;
}
constexpr int kNcclUint8 = 1;  // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1 (nccl.h / rccl.h, stable ABI)
}  // namespace

namespace snarkv {

static void shard(size_t n, int g, int world, size_t* lo, size_t* hi) {  // msm.rs:322: chunk = ceil(n / threads)
  size_t chunk = (n + (size_t)world - 1) / (size_t)world;
  *lo = std::min((size_t)g * chunk, n);
  *hi = std::min(*lo + chunk, n);
}

// rank g's stream waits for everything rank src has enqueued so far
static int wait_for(snarkv_mgpu* mg, int g, int src) {
  SNARKV_HIP(hipSetDevice(mg->ctx[src]->device));
  SNARKV_HIP(hipEventRecord(mg->done[src], mg->ctx[src]->stream));
  SNARKV_HIP(hipSetDevice(mg->ctx[g]->device));
  SNARKV_HIP(hipStreamWaitEvent(mg->ctx[g]->stream, mg->done[src], 0));
  return SNARKV_OK;
}

// partials of all ranks -> fold -> the 64-byte result on EVERY rank's device (mg->d_result) and in out64
static int gather_fold(snarkv_mgpu* mg, uint8_t out64[64]) {
  const int world = (int)mg->ctx.size();
  snarkv_ctx* c0 = mg->ctx[0];
  if (mg->transport == SNARKV_MGPU_TRANSPORT_RCCL) {
    RcclApi* nc = rccl_api();
    if (!nc || (int)mg->comms.size() != world) {
      set_last_error("mgpu: RCCL transport selected but not initialised");
      return SNARKV_ERR_DEVICE;
    }
    // one grouped all-gather: rank g contributes its 144-byte partial, every rank receives [rank][144]
    int rc = nc->GroupStart();
    hipError_t herr = hipSuccess;  // (no early return inside the group: GroupEnd always closes it)
    for (int g = 0; g < world && rc == 0 && herr == hipSuccess; ++g) {
      herr = hipSetDevice(mg->ctx[g]->device);
      if (herr != hipSuccess) break;
      rc = nc->AllGather(mg->d_part[g], mg->d_gather_all[g], SNARKV_G1_PARTIAL_BYTES, kNcclUint8, mg->comms[g], mg->ctx[g]->stream);
    }
    int rc2 = nc->GroupEnd();
    if (herr != hipSuccess) {
      set_last_error("mgpu: hipSetDevice inside the all-gather group: %s", hipGetErrorString(herr));
      return SNARKV_ERR_DEVICE;
    }
    if (rc != 0 || rc2 != 0) {
      set_last_error("mgpu: ncclAllGather failed: %s", nc->GetErrorString ? nc->GetErrorString(rc ? rc : rc2) : "?");
      return SNARKV_ERR_DEVICE;
    }
    for (int g = 0; g < world; ++g) {  // every rank folds all partials itself: all ranks hold the identical result
      SNARKV_HIP(hipSetDevice(mg->ctx[g]->device));
      SNARKV_TRY(launch_fold_partials(mg->ctx[g], mg->d_gather_all[g], (size_t)world, mg->d_result[g]));
    }
  } else {
    for (int g = 0; g < world; ++g) {
      snarkv_ctx* c = mg->ctx[g];
      SNARKV_HIP(hipSetDevice(c->device));
      // the copy is enqueued on the PRODUCING rank's stream (ordered after its kernels), then rank 0 waits for it
      SNARKV_HIP(hipMemcpyPeerAsync((char*)mg->d_gather[0] + (size_t)g * SNARKV_G1_PARTIAL_BYTES, c0->device, mg->d_part[g],
                                    c->device, SNARKV_G1_PARTIAL_BYTES, c->stream));
      if (g != 0) SNARKV_TRY(wait_for(mg, 0, g));
    }
    SNARKV_HIP(hipSetDevice(c0->device));
    SNARKV_TRY(launch_fold_partials(c0, mg->d_gather[0], (size_t)world, mg->d_result[0]));
    // ... and back: the folded 64 bytes to every other rank's device (ordered after the fold: on rank 0's stream)
    for (int g = 1; g < world; ++g)
      SNARKV_HIP(hipMemcpyPeerAsync(mg->d_result[g], mg->ctx[g]->device, mg->d_result[0], c0->device, 64, c0->stream));
  }
  SNARKV_HIP(hipSetDevice(c0->device));
  SNARKV_HIP(hipMemcpyAsync(out64, mg->d_result[0], 64, hipMemcpyDeviceToHost, c0->stream));
  SNARKV_HIP(hipStreamSynchronize(c0->stream));
  if (mg->transport == SNARKV_MGPU_TRANSPORT_RCCL)
    for (int g = 1; g < world; ++g) {  // the other ranks' folds ran on their own streams
      SNARKV_HIP(hipSetDevice(mg->ctx[g]->device));
      SNARKV_HIP(hipStreamSynchronize(mg->ctx[g]->stream));
    }
  return SNARKV_OK;
}

static int msm_point_sharded(snarkv_mgpu* mg, const void* const* d_s, const void* const* d_p, const size_t* counts,
                             int window_bits, uint8_t out64[64]) {
  const int world = (int)mg->ctx.size();
  for (int g = 0; g < world; ++g) {
    snarkv_ctx* c = mg->ctx[g];
    SNARKV_HIP(hipSetDevice(c->device));
    if (counts[g] == 0) {  // ceil chunking leaves trailing ranks empty when n is small: the identity (ZZ = 0)
      SNARKV_HIP(hipMemsetAsync(mg->d_part[g], 0, SNARKV_G1_PARTIAL_BYTES, c->stream));
      continue;
    }
    SNARKV_TRY(launch_msm_pippenger_auto(c, d_s[g], d_p[g], counts[g], window_bits, mg->d_part[g], true));
  }
  return gather_fold(mg, out64);
}

static int msm_bucket_sharded(snarkv_mgpu* mg, const void* const* d_s, const void* const* d_p, const size_t* counts,
                              size_t n_total, int window_bits, uint8_t out64[64]) {
  const int world = (int)mg->ctx.size();
  uint32_t c = 0, windows = 0, bpw = 0;
  SNARKV_TRY(pip_geometry(n_total, window_bits, &c, &windows, &bpw));  // ONE geometry for all ranks: that of the total
  const size_t wbytes = (size_t)bpw * SNARKV_G1_PARTIAL_BYTES, grid_bytes = wbytes * windows;
  std::vector<void*> grid(world), recv(world);
  std::vector<size_t> w0(world), w1(world);
  for (int g = 0; g < world; ++g) {
    shard(windows, g, world, &w0[g], &w1[g]);
    snarkv_ctx* cx = mg->ctx[g];
    SNARKV_HIP(hipSetDevice(cx->device));
    SNARKV_TRY(ctx_reserve(cx, SLOT_MGPU_GRID, grid_bytes, &grid[g]));
    SNARKV_TRY(ctx_reserve(cx, SLOT_MGPU_RECV, std::max<size_t>(16, (w1[g] - w0[g]) * wbytes), &recv[g]));
    if (counts[g] == 0) SNARKV_HIP(hipMemsetAsync(grid[g], 0, grid_bytes, cx->stream));
    else SNARKV_TRY(launch_msm_pippenger(cx, d_s[g], d_p[g], counts[g], (int)c, nullptr, false, grid[g]));
    SNARKV_HIP(hipEventRecord(mg->filled[g], cx->stream));
  }
  // exchange by window range: owner g receives every other rank's copy of windows [w0, w1) and adds it to its own
  for (int g = 0; g < world; ++g) {
    if (w1[g] == w0[g]) continue;
    snarkv_ctx* cx = mg->ctx[g];
    const size_t own = (w1[g] - w0[g]) * wbytes;
    for (int src = 0; src < world; ++src) {
      if (src == g) continue;
      snarkv_ctx* cs = mg->ctx[src];
      SNARKV_HIP(hipSetDevice(cx->device));
      SNARKV_HIP(hipStreamWaitEvent(cx->stream, mg->filled[src], 0));  // src's grid is complete
      SNARKV_HIP(hipMemcpyPeerAsync(recv[g], cx->device, (const char*)grid[src] + w0[g] * wbytes, cs->device, own, cx->stream));
      SNARKV_TRY(launch_buckets_add(cx, (char*)grid[g] + w0[g] * wbytes, recv[g], own / SNARKV_G1_PARTIAL_BYTES));
    }
  }
  for (int g = 0; g < world; ++g) {
    snarkv_ctx* cx = mg->ctx[g];
    SNARKV_HIP(hipSetDevice(cx->device));
    if (w1[g] == w0[g]) {  // more ranks than windows: the identity
      SNARKV_HIP(hipMemsetAsync(mg->d_part[g], 0, SNARKV_G1_PARTIAL_BYTES, cx->stream));
      continue;
    }
    SNARKV_TRY(launch_buckets_reduce(cx, (const char*)grid[g] + w0[g] * wbytes, c, (uint32_t)w0[g], (uint32_t)(w1[g] - w0[g]),
                                     mg->d_part[g]));
  }
  // a grid must not be overwritten by this rank's NEXT call while a peer still copies from it: all ranks drain here
  int rc = gather_fold(mg, out64);
  for (int g = 0; g < world; ++g) {
    (void)hipSetDevice(mg->ctx[g]->device);
    (void)hipStreamSynchronize(mg->ctx[g]->stream);
  }
  return rc;
}

// [rank][job][144 B] -> [job][rank][144 B]: one 4-byte word per lane (the gathered partials of a batch, before the folds)
__global__ void k_transpose_partials(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t world, uint32_t jobs) {
  constexpr uint32_t W = SNARKV_G1_PARTIAL_BYTES / 4;
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= world * jobs * W) return;
  const uint32_t w = id % W, j = (id / W) % jobs, g = id / (W * jobs);
  out[((size_t)j * world + g) * W + w] = in[id];
}

static int many_reserve(snarkv_mgpu* mg, size_t jobs) {
  if (jobs <= mg->many_cap) return SNARKV_OK;
  const int world = (int)mg->ctx.size();
  for (int g = 0; g < world; ++g) {  // queued work may still read the old buffers
    SNARKV_HIP(hipSetDevice(mg->ctx[g]->device));
    SNARKV_HIP(hipStreamSynchronize(mg->ctx[g]->stream));
  }
  auto regrow = [&](std::vector<void*>& v, size_t bytes) -> int {
    v.resize(world, nullptr);
    for (int g = 0; g < world; ++g) {
      SNARKV_HIP(hipSetDevice(mg->ctx[g]->device));
      if (v[g]) SNARKV_HIP(hipFree(v[g]));
      v[g] = nullptr;
      SNARKV_HIP(hipMalloc(&v[g], bytes));
    }
    return SNARKV_OK;
  };
  const size_t cap = std::max<size_t>(jobs, 32);
  mg->many_cap = 0;
  SNARKV_TRY(regrow(mg->d_parts_many, cap * SNARKV_G1_PARTIAL_BYTES));
  SNARKV_TRY(regrow(mg->d_gather_many, cap * world * SNARKV_G1_PARTIAL_BYTES));
  SNARKV_TRY(regrow(mg->d_byjob_many, cap * world * SNARKV_G1_PARTIAL_BYTES));
  SNARKV_TRY(regrow(mg->d_results_many, cap * 64));
  mg->many_cap = cap;
  return SNARKV_OK;
}

// rank g's K partials (its shard of every job): one batch call when every shard is non-empty, else job by job with the
// identity (ZZ = 0) in the empty slots -- an empty shard must never reach the batch call (SNARKV_ERR_EMPTY)
static int many_partials_rank(snarkv_mgpu* mg, int g, size_t jobs, const void* const* d_s, const void* const* d_p,
                              const size_t* counts, int window_bits) {
  snarkv_ctx* c = mg->ctx[g];
  SNARKV_HIP(hipSetDevice(c->device));
  bool all_live = true;
  for (size_t j = 0; j < jobs; ++j) all_live = all_live && counts[j] != 0;
  if (all_live) return launch_msm_pippenger_many(c, jobs, d_s, d_p, counts, window_bits, mg->d_parts_many[g], true);
  SNARKV_HIP(hipMemsetAsync(mg->d_parts_many[g], 0, jobs * SNARKV_G1_PARTIAL_BYTES, c->stream));
  for (size_t j = 0; j < jobs; ++j)
    if (counts[j])
      SNARKV_TRY(launch_msm_pippenger_auto(c, d_s[j], d_p[j], counts[j], window_bits,
                                           (uint8_t*)mg->d_parts_many[g] + j * SNARKV_G1_PARTIAL_BYTES, true));
  return SNARKV_OK;
}

static int transpose_fold(snarkv_mgpu* mg, int g, size_t jobs) {
  const int world = (int)mg->ctx.size();
  snarkv_ctx* c = mg->ctx[g];
  SNARKV_HIP(hipSetDevice(c->device));
  const uint32_t words = (uint32_t)(world * jobs * (SNARKV_G1_PARTIAL_BYTES / 4));
  hipLaunchKernelGGL(k_transpose_partials, dim3((words + 255) / 256), dim3(256), 0, c->stream,
                     (const uint32_t*)mg->d_gather_many[g], (uint32_t*)mg->d_byjob_many[g], (uint32_t)world, (uint32_t)jobs);
  SNARKV_HIP(hipGetLastError());
  return launch_fold_partials_many(c, mg->d_byjob_many[g], (size_t)world, jobs, mg->d_results_many[g]);
}

// K jobs, each sharded over the ranks, ONE exchange for the whole batch (K x 144 B per rank): the single-process form of
// snark-verifier_amd/distributed.py::gpu_sharded_msm_batch.  The ranks' batches are enqueued by one host thread each --
// a 20-job batch is ~400 launches per rank, and one thread walking eight ranks would start the last of them several
// milliseconds after the first.
static int many_exchange_and_fold(snarkv_mgpu* mg, size_t jobs, uint8_t* out64s);

// nothing of a failed batch stays queued behind the caller's back: the partial kernels (~400 launches per rank) read the
// caller's d_scalars / d_points, and a caller that frees or reuses them after an error return must not race with the device
static void drain_all_ranks(snarkv_mgpu* mg) {
  for (size_t h = 0; h < mg->ctx.size(); ++h) {
    (void)hipSetDevice(mg->ctx[h]->device);
    (void)hipStreamSynchronize(mg->ctx[h]->stream);
  }
  (void)hipGetLastError();
}

static int msm_many_point_sharded(snarkv_mgpu* mg, size_t jobs, const void* const* d_s, const void* const* d_p,
                                  const size_t* counts, int window_bits, uint8_t* out64s) {
  const int world = (int)mg->ctx.size();
  SNARKV_TRY(many_reserve(mg, jobs));
  std::vector<int> rcs(world, SNARKV_OK);
  std::vector<std::string> errs(world);
  auto run_rank = [&](int g) {
    rcs[g] = many_partials_rank(mg, g, jobs, d_s + (size_t)g * jobs, d_p + (size_t)g * jobs, counts + (size_t)g * jobs, window_bits);
    if (rcs[g] < 0) errs[g] = snarkv_last_error();  // (thread-local: carried to the caller's thread below)
  };
  if (world == 1) {
    run_rank(0);
  } else {
    std::vector<std::thread> th;
    try {
      for (int g = 0; g < world; ++g) th.emplace_back(run_rank, g);
    } catch (const std::exception& e) {  // (no thread to be had: the ranks that have none run here, one after the other)
      for (int g = (int)th.size(); g < world; ++g) run_rank(g);
    }
    for (auto& t : th) t.join();
  }
  for (int g = 0; g < world; ++g)
    if (rcs[g] < 0) {
      set_last_error("mgpu rank %d: %s", g, errs[g].c_str());
      drain_all_ranks(mg);
      return rcs[g];
    }
  // every error exit of the exchange (RCCL not initialised, hipSetDevice / AllGather / GroupEnd, a peer copy) leaves through
  // the same drain as a failed rank
  int rc = many_exchange_and_fold(mg, jobs, out64s);
  if (rc < 0) drain_all_ranks(mg);
  return rc;
}

static int many_exchange_and_fold(snarkv_mgpu* mg, size_t jobs, uint8_t* out64s) {
  const int world = (int)mg->ctx.size();
  const size_t row = jobs * SNARKV_G1_PARTIAL_BYTES;
  snarkv_ctx* c0 = mg->ctx[0];
  if (mg->transport == SNARKV_MGPU_TRANSPORT_RCCL) {
    RcclApi* nc = rccl_api();
    if (!nc || (int)mg->comms.size() != world) {
      set_last_error("mgpu: RCCL transport selected but not initialised");
      return SNARKV_ERR_DEVICE;
    }
    int rc = nc->GroupStart();
    hipError_t herr = hipSuccess;
    for (int g = 0; g < world && rc == 0 && herr == hipSuccess; ++g) {
      herr = hipSetDevice(mg->ctx[g]->device);
      if (herr != hipSuccess) break;
      rc = nc->AllGather(mg->d_parts_many[g], mg->d_gather_many[g], row, kNcclUint8, mg->comms[g], mg->ctx[g]->stream);
    }
    int rc2 = nc->GroupEnd();
    if (herr != hipSuccess) {
      set_last_error("mgpu: hipSetDevice inside the all-gather group: %s", hipGetErrorString(herr));
      return SNARKV_ERR_DEVICE;
    }
    if (rc != 0 || rc2 != 0) {
      set_last_error("mgpu: ncclAllGather failed: %s", nc->GetErrorString ? nc->GetErrorString(rc ? rc : rc2) : "?");
      return SNARKV_ERR_DEVICE;
    }
    for (int g = 0; g < world; ++g) SNARKV_TRY(transpose_fold(mg, g, jobs));  // every rank folds: identical results everywhere
  } else {
    for (int g = 0; g < world; ++g) {
      snarkv_ctx* c = mg->ctx[g];
      SNARKV_HIP(hipSetDevice(c->device));
      SNARKV_HIP(hipMemcpyPeerAsync((char*)mg->d_gather_many[0] + (size_t)g * row, c0->device, mg->d_parts_many[g], c->device, row,
                                    c->stream));
      if (g != 0) SNARKV_TRY(wait_for(mg, 0, g));
    }
    SNARKV_TRY(transpose_fold(mg, 0, jobs));
    for (int g = 1; g < world; ++g)
      SNARKV_HIP(hipMemcpyPeerAsync(mg->d_results_many[g], mg->ctx[g]->device, mg->d_results_many[0], c0->device, jobs * 64, c0->stream));
  }
  SNARKV_HIP(hipSetDevice(c0->device));
  SNARKV_HIP(hipMemcpyAsync(out64s, mg->d_results_many[0], jobs * 64, hipMemcpyDeviceToHost, c0->stream));
  SNARKV_HIP(hipStreamSynchronize(c0->stream));
  if (mg->transport == SNARKV_MGPU_TRANSPORT_RCCL)
    for (int g = 1; g < world; ++g) {
      SNARKV_HIP(hipSetDevice(mg->ctx[g]->device));
      SNARKV_HIP(hipStreamSynchronize(mg->ctx[g]->stream));
    }
  return SNARKV_OK;
}

}  // namespace snarkv

using namespace snarkv;

extern "C" {

int snarkv_mgpu_create(const int* devices, int n, snarkv_mgpu** out) {
  if (!devices || !out || n <= 0 || n > 64) return SNARKV_ERR_ARG;
  *out = nullptr;
  snarkv_mgpu* mg = new snarkv_mgpu();
  auto fail = [&](int rc) {
    snarkv_mgpu_destroy(mg);
    return rc;
  };
  for (int g = 0; g < n; ++g) {
    snarkv_ctx* c = nullptr;
    int rc = snarkv_ctx_create(devices[g], nullptr, &c);
    if (rc < 0) return fail(rc);
    mg->ctx.push_back(c);
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(SNARKV_ERR_DEVICE);
    mg->done.push_back(ev);
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(SNARKV_ERR_DEVICE);
    mg->filled.push_back(ev);
    void* p = nullptr;
    if (hipMalloc(&p, SNARKV_G1_PARTIAL_BYTES) != hipSuccess) return fail(SNARKV_ERR_DEVICE);
    mg->d_part.push_back(p);
  }
  for (int g = 0; g < n; ++g) {
    (void)hipSetDevice(devices[g]);
    void* p = nullptr;
    if (hipMalloc(&p, 64) != hipSuccess) return fail(SNARKV_ERR_DEVICE);
    mg->d_result.push_back(p);
  }
  (void)hipSetDevice(devices[0]);
  void* gbuf = nullptr;
  if (hipMalloc(&gbuf, (size_t)n * SNARKV_G1_PARTIAL_BYTES) != hipSuccess) return fail(SNARKV_ERR_DEVICE);
  mg->d_gather.push_back(gbuf);
  // Direct xGMI access between DISTINCT devices where the platform allows it.  Peer copies work either way (the runtime
  // stages through the host when it must), so a pair that cannot be enabled is not fatal -- but it is not silent either:
  // the counts are kept (snarkv_mgpu_peer_access) and the last refusal is left in snarkv_last_error().
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      if (devices[a] == devices[b]) continue;
      bool seen = false;  // count a directed device pair once, however many ranks share the devices
      for (int a2 = 0; a2 <= a && !seen; ++a2)
        for (int b2 = 0; b2 < (a2 == a ? b : n) && !seen; ++b2) seen = devices[a2] == devices[a] && devices[b2] == devices[b];
      if (seen) continue;
      int can = 0;
      hipError_t e = hipDeviceCanAccessPeer(&can, devices[a], devices[b]);
      if (e != hipSuccess || !can) {
        ++mg->peer_unavailable;
        set_last_error("mgpu: device %d cannot access device %d directly (%s): copies between them are staged by the runtime",
                       devices[a], devices[b], e != hipSuccess ? hipGetErrorString(e) : "hipDeviceCanAccessPeer = 0");
        (void)hipGetLastError();
        continue;
      }
      (void)hipSetDevice(devices[a]);
      e = hipDeviceEnablePeerAccess(devices[b], 0);
      if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) {
        ++mg->peer_enabled;
      } else {
        ++mg->peer_failed;
        set_last_error("mgpu: hipDeviceEnablePeerAccess(%d -> %d) failed: %s", devices[a], devices[b], hipGetErrorString(e));
      }
      (void)hipGetLastError();
    }
  *out = mg;
  return SNARKV_OK;
}

int snarkv_mgpu_peer_access(const snarkv_mgpu* mg, int* enabled, int* unavailable, int* failed) {
  if (!mg) return SNARKV_ERR_ARG;
  if (enabled) *enabled = mg->peer_enabled;
  if (unavailable) *unavailable = mg->peer_unavailable;
  if (failed) *failed = mg->peer_failed;
  return SNARKV_OK;
}

int snarkv_mgpu_set_transport(snarkv_mgpu* mg, int transport) {
  if (!mg || (transport != SNARKV_MGPU_TRANSPORT_PEER_COPY && transport != SNARKV_MGPU_TRANSPORT_RCCL)) return SNARKV_ERR_ARG;
  if (transport == SNARKV_MGPU_TRANSPORT_RCCL && mg->comms.empty()) {
    const int world = (int)mg->ctx.size();
    RcclApi* nc = rccl_api();
    if (!nc) {
      set_last_error("mgpu: librccl could not be loaded (%s)", g_rccl_why.empty() ? "symbols missing" : g_rccl_why.c_str());
      return SNARKV_ERR_DEVICE;
    }
    std::vector<int> devs(world);
    for (int g = 0; g < world; ++g) devs[g] = mg->ctx[g]->device;
    for (int a = 0; a < world; ++a)
      for (int b = a + 1; b < world; ++b)
        if (devs[a] == devs[b]) {
          set_last_error("mgpu: the RCCL transport needs distinct devices (device %d is listed twice)", devs[a]);
          return SNARKV_ERR_ARG;
        }
    std::vector<void*> comms(world, nullptr);
    int rc = nc->CommInitAll(comms.data(), world, devs.data());
    if (rc != 0) {
      set_last_error("mgpu: ncclCommInitAll over %d devices failed: %s", world, nc->GetErrorString ? nc->GetErrorString(rc) : "?");
      return SNARKV_ERR_DEVICE;
    }
    mg->comms = comms;
    for (int g = 0; g < world; ++g) {
      SNARKV_HIP(hipSetDevice(devs[g]));
      void* p = nullptr;
      SNARKV_HIP(hipMalloc(&p, (size_t)world * SNARKV_G1_PARTIAL_BYTES));
      mg->d_gather_all.push_back(p);
    }
  }
  mg->transport = transport;
  return SNARKV_OK;
}

const void* snarkv_mgpu_result_dev(const snarkv_mgpu* mg, int rank) {
  return (mg && rank >= 0 && rank < (int)mg->d_result.size()) ? mg->d_result[rank] : nullptr;
}

void snarkv_mgpu_destroy(snarkv_mgpu* mg) {
  if (!mg) return;
  for (size_t g = 0; g < mg->ctx.size(); ++g) {
    (void)hipSetDevice(mg->ctx[g]->device);
    (void)hipStreamSynchronize(mg->ctx[g]->stream);
  }
  for (auto* d : mg->dk) snarkv_dk_destroy(d);
  for (size_t g = 0; g < mg->d_part.size(); ++g) {
    (void)hipSetDevice(mg->ctx[g]->device);
    (void)hipFree(mg->d_part[g]);
  }
  for (size_t g = 0; g < mg->d_result.size(); ++g) {
    (void)hipSetDevice(mg->ctx[g]->device);
    (void)hipFree(mg->d_result[g]);
  }
  for (size_t g = 0; g < mg->d_gather_all.size(); ++g) {
    (void)hipSetDevice(mg->ctx[g]->device);
    (void)hipFree(mg->d_gather_all[g]);
  }
  for (auto* v : {&mg->d_parts_many, &mg->d_gather_many, &mg->d_byjob_many, &mg->d_results_many})
    for (size_t g = 0; g < v->size(); ++g) {
      (void)hipSetDevice(mg->ctx[g]->device);
      if ((*v)[g]) (void)hipFree((*v)[g]);
    }
  if (RcclApi* nc = mg->comms.empty() ? nullptr : rccl_api())
    for (void* c : mg->comms)
      if (c) (void)nc->CommDestroy(c);
  if (!mg->d_gather.empty()) {
    (void)hipSetDevice(mg->ctx[0]->device);
    (void)hipFree(mg->d_gather[0]);
  }
  for (auto ev : mg->done) (void)hipEventDestroy(ev);
  for (auto ev : mg->filled) (void)hipEventDestroy(ev);
  for (auto* c : mg->ctx) snarkv_ctx_destroy(c);
  delete mg;
}

int snarkv_mgpu_size(const snarkv_mgpu* mg) { return mg ? (int)mg->ctx.size() : SNARKV_ERR_ARG; }

snarkv_ctx* snarkv_mgpu_ctx(snarkv_mgpu* mg, int rank) {
  return (mg && rank >= 0 && rank < (int)mg->ctx.size()) ? mg->ctx[rank] : nullptr;
}

int snarkv_mgpu_shard(const snarkv_mgpu* mg, size_t n_total, int rank, size_t* lo, size_t* hi) {
  if (!mg || !lo || !hi || rank < 0 || rank >= (int)mg->ctx.size()) return SNARKV_ERR_ARG;
  shard(n_total, rank, (int)mg->ctx.size(), lo, hi);
  return SNARKV_OK;
}

int snarkv_g1_msm_pippenger_mgpu_dev(snarkv_mgpu* mg, const void* const* d_scalars32, const void* const* d_points64,
                                     const size_t* counts, int window_bits, int variant, uint8_t out64[64]) {
  if (!mg || !d_scalars32 || !d_points64 || !counts || !out64 || variant < 0 || variant > 1) return SNARKV_ERR_ARG;
  const int world = (int)mg->ctx.size();
  size_t total = 0;
  for (int g = 0; g < world; ++g) {
    if (counts[g] && (!d_scalars32[g] || !d_points64[g])) return SNARKV_ERR_ARG;
    total += counts[g];
  }
  if (total == 0) return SNARKV_ERR_EMPTY;  // reference panics: msm.rs:265
  return variant == 0 ? msm_point_sharded(mg, d_scalars32, d_points64, counts, window_bits, out64)
                      : msm_bucket_sharded(mg, d_scalars32, d_points64, counts, total, window_bits, out64);
}

int snarkv_g1_msm_pippenger_many_mgpu_dev(snarkv_mgpu* mg, size_t jobs, const void* const* d_scalars32,
                                          const void* const* d_points64, const size_t* counts, int window_bits,
                                          uint8_t* out64s) {
  if (!mg || !d_scalars32 || !d_points64 || !counts || !out64s) return SNARKV_ERR_ARG;
  if (jobs == 0) return SNARKV_ERR_EMPTY;
  if (jobs > 65536) return SNARKV_ERR_LENGTH;  // (the exchange buffers and the transposing kernel index jobs x ranks x 36 words in 32 bits)
  const int world = (int)mg->ctx.size();
  for (size_t j = 0; j < jobs; ++j) {
    size_t total = 0;
    for (int g = 0; g < world; ++g) {
      const size_t i = (size_t)g * jobs + j;
      if (counts[i] && (!d_scalars32[i] || !d_points64[i])) return SNARKV_ERR_ARG;
      total += counts[i];
    }
    if (total == 0) return SNARKV_ERR_EMPTY;  // reference panics on an empty MSM: msm.rs:265
  }
  return msm_many_point_sharded(mg, jobs, d_scalars32, d_points64, counts, window_bits, out64s);
}

const void* snarkv_mgpu_results_many_dev(const snarkv_mgpu* mg, int rank) {
  return (mg && rank >= 0 && rank < (int)mg->d_results_many.size()) ? mg->d_results_many[rank] : nullptr;
}

int snarkv_g1_msm_pippenger_mgpu(snarkv_mgpu* mg, const uint8_t* scalars32, const uint8_t* points64, size_t n, int variant,
                                 uint8_t out64[64]) {
  if (!mg || !scalars32 || !points64 || !out64) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_ERR_EMPTY;
  const int world = (int)mg->ctx.size();
  std::vector<const void*> ds(world, nullptr), dp(world, nullptr);
  std::vector<size_t> counts(world, 0);
  for (int g = 0; g < world; ++g) {
    size_t lo, hi;
    shard(n, g, world, &lo, &hi);
    counts[g] = hi - lo;
    if (hi == lo) continue;
    snarkv_ctx* c = mg->ctx[g];
    SNARKV_HIP(hipSetDevice(c->device));
    void *s, *p;
    SNARKV_TRY(ctx_reserve(c, SLOT_IN_SCALARS, (hi - lo) * 32, &s));
    SNARKV_TRY(ctx_reserve(c, SLOT_IN_POINTS, (hi - lo) * 64, &p));
    SNARKV_HIP(hipMemcpyAsync(s, scalars32 + 32 * lo, (hi - lo) * 32, hipMemcpyHostToDevice, c->stream));
    SNARKV_HIP(hipMemcpyAsync(p, points64 + 64 * lo, (hi - lo) * 64, hipMemcpyHostToDevice, c->stream));
    ds[g] = s;
    dp[g] = p;
  }
  return snarkv_g1_msm_pippenger_mgpu_dev(mg, ds.data(), dp.data(), counts.data(), 0, variant, out64);
}

int snarkv_kzg_decide_batch_mgpu(snarkv_mgpu* mg, const uint8_t g1_64[64], const uint8_t g2_128[128],
                                 const uint8_t s_g2_128[128], const uint8_t* accs128, size_t m, uint8_t* ok) {
  if (!mg || !g1_64 || !g2_128 || !s_g2_128 || (m && (!accs128 || !ok))) return SNARKV_ERR_ARG;
  if (m == 0) return 1;  // decide_all over an empty list is Ok(()) (decider.rs:84-93)
  const int world = (int)mg->ctx.size();
  uint8_t key[320];
  memcpy(key, g1_64, 64);
  memcpy(key + 64, g2_128, 128);
  memcpy(key + 192, s_g2_128, 128);
  if (!mg->dk_valid || memcmp(key, mg->dk_bytes, 320) != 0) {  // G2 line tables: once per key per rank
    for (auto* d : mg->dk) snarkv_dk_destroy(d);
    mg->dk.clear();
    mg->dk_valid = false;
    for (int g = 0; g < world; ++g) {
      snarkv_dk* d = nullptr;
      SNARKV_TRY(snarkv_dk_create(mg->ctx[g], g1_64, g2_128, s_g2_128, 0, &d));
      mg->dk.push_back(d);
    }
    memcpy(mg->dk_bytes, key, 320);
    mg->dk_valid = true;
  }
  std::vector<void*> d_ok(world, nullptr);
  std::vector<size_t> lo(world), hi(world);
  for (int g = 0; g < world; ++g) {
    shard(m, g, world, &lo[g], &hi[g]);
    if (hi[g] == lo[g]) continue;
    snarkv_ctx* c = mg->ctx[g];
    SNARKV_HIP(hipSetDevice(c->device));
    void* d_a;
    SNARKV_TRY(ctx_reserve(c, SLOT_IN_POINTS, (hi[g] - lo[g]) * 128, &d_a));
    SNARKV_TRY(ctx_reserve(c, SLOT_OUT, hi[g] - lo[g], &d_ok[g]));
    SNARKV_HIP(hipMemcpyAsync(d_a, accs128 + 128 * lo[g], (hi[g] - lo[g]) * 128, hipMemcpyHostToDevice, c->stream));
    SNARKV_TRY(launch_decide(c, mg->dk[g]->d_prep, d_a, hi[g] - lo[g], d_ok[g], nullptr));
    SNARKV_HIP(hipMemcpyAsync(ok + lo[g], d_ok[g], hi[g] - lo[g], hipMemcpyDeviceToHost, c->stream));
  }
  int all = 1;
  for (int g = 0; g < world; ++g) {
    if (hi[g] == lo[g]) continue;
    SNARKV_HIP(hipSetDevice(mg->ctx[g]->device));
    SNARKV_HIP(hipStreamSynchronize(mg->ctx[g]->stream));
  }
  for (size_t i = 0; i < m; ++i) all &= ok[i] ? 1 : 0;
  return all;
}

}  // extern "C"
