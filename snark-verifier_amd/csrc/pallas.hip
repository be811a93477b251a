// The pasta build of the curve-generic part of the library -> libsnarkv_pallas.so
// (include/snarkv_pallas.h).  The reference's IPA layer is generic over `C: CurveAffine`
// and its own tests run it on pallas (snark-verifier/src/pcs/ipa.rs:434-466,
// pcs/ipa/accumulation.rs:240-290), whose only device-worthy work is
// `util::msm::multi_scalar_multiplication` (msm.rs:308-343: `IpaProvingKey::commit`,
// pcs/ipa.rs:220-229, and `IpaAs::decide`, pcs/ipa/decider.rs:47-55).
//
// Same sources as the BN254 library -- fq29.h / fr29.h / g1_29.h / glv.h / msm_pippenger.hip /
// msm_naive.hip / ipa.hip -- compiled with
//   -DSNARKV_CURVE_PALLAS   pallas_consts.h: p, r, b = 5 (curve_consts.h)
//   -Dsnarkv=snarkv_pallas  the C++ namespace, so both libraries can live in one process
// msm_naive.hip (the segmented small-MSM kernels behind `NativeLoader::multi_scalar_multiplication`,
// loader/native.rs:61-71) comes along; pallas has the same kind of endomorphism as BN254 (j = 0), so
// the GLV split stays on, with pallas' lattice.  There is no pairing, no KZG decider and no transcript.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include "ctx.hpp"
#include "../../include/snarkv_pallas.h"

#include "ctx_impl.inc"

using namespace snarkv;

extern "C" {

int snarkv_pallas_g1_msm_pippenger_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64, size_t n,
                                       int window_bits, void* d_out64) {
  if (!ctx || !d_scalars32 || !d_points64 || !d_out64) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_ERR_EMPTY;
  SNARKV_HIP(hipSetDevice(ctx->device));
  return launch_msm_pippenger(ctx, d_scalars32, d_points64, n, window_bits, d_out64, false);
}

int snarkv_pallas_g1_msm_pippenger(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                                   uint8_t out64[64]) {
  if (!ctx || !scalars32 || !points64 || !out64) return SNARKV_ERR_ARG;
  if (n == 0) return SNARKV_ERR_EMPTY;  // reference: `scalars[0]` panics (msm.rs:265)
  SNARKV_HIP(hipSetDevice(ctx->device));
  void *d_s, *d_p, *d_o;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_SCALARS, n * 32, &d_s));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_POINTS, n * 64, &d_p));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, 64, &d_o));
  SNARKV_HIP(hipMemcpyAsync(d_s, scalars32, n * 32, hipMemcpyHostToDevice, ctx->stream));
  SNARKV_HIP(hipMemcpyAsync(d_p, points64, n * 64, hipMemcpyHostToDevice, ctx->stream));
  SNARKV_TRY(launch_msm_pippenger(ctx, d_s, d_p, n, 0, d_o, false));
  SNARKV_HIP(hipMemcpyAsync(out64, d_o, 64, hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  return SNARKV_OK;
}

// `NativeLoader::multi_scalar_multiplication` (loader/native.rs:61-71) for C = pallas::Affine, n_msm
// independent MSMs in one launch (segment k = terms offsets[k] .. offsets[k+1])
int snarkv_pallas_g1_msm_batched(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64,
                                 const uint32_t* offsets, size_t n_msm, uint32_t flags, uint8_t* out) {
  if (!ctx || !scalars32 || !points64 || !offsets || !out) return SNARKV_ERR_ARG;
  if (n_msm == 0) return SNARKV_ERR_EMPTY;
  if (offsets[0] != 0) return SNARKV_ERR_LENGTH;
  for (size_t k = 0; k < n_msm; ++k) {
    if (offsets[k + 1] < offsets[k]) return SNARKV_ERR_LENGTH;
    if (offsets[k + 1] == offsets[k]) return SNARKV_ERR_EMPTY;  // reference panics: native.rs:69
  }
  const size_t n = offsets[n_msm];
  SNARKV_HIP(hipSetDevice(ctx->device));
  void *d_s, *d_p, *d_o, *d_out;
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_SCALARS, n * 32, &d_s));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_POINTS, n * 64, &d_p));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_IN_OFFSETS, (n_msm + 1) * 4, &d_o));
  SNARKV_TRY(ctx_reserve(ctx, SLOT_OUT, n_msm * 64, &d_out));
  SNARKV_HIP(hipMemcpyAsync(d_s, scalars32, n * 32, hipMemcpyHostToDevice, ctx->stream));
  SNARKV_HIP(hipMemcpyAsync(d_p, points64, n * 64, hipMemcpyHostToDevice, ctx->stream));
  SNARKV_HIP(hipMemcpyAsync(d_o, offsets, (n_msm + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (flags & SNARKV_FLAG_VALIDATE) {
    int bad = 0;
    SNARKV_TRY(launch_validate(ctx, d_s, d_p, n, &bad));
    if (bad) {
      set_last_error("%d of %zu inputs are non-canonical or off-curve", bad, n);
      return SNARKV_ERR_ENCODING;
    }
  }
  SNARKV_TRY(launch_msm_batched(ctx, d_s, d_p, d_o, n_msm, n, d_out));
  SNARKV_HIP(hipMemcpyAsync(out, d_out, n_msm * 64, hipMemcpyDeviceToHost, ctx->stream));
  SNARKV_HIP(hipStreamSynchronize(ctx->stream));
  return SNARKV_OK;
}

int snarkv_pallas_g1_msm_naive(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                               uint32_t flags, uint8_t out64[64]) {
  if (n == 0) return SNARKV_ERR_EMPTY;
  if (n > 0xFFFFFFFFull) return SNARKV_ERR_LENGTH;
  uint32_t offsets[2] = {0, (uint32_t)n};
  return snarkv_pallas_g1_msm_batched(ctx, scalars32, points64, offsets, 1, flags, out64);
}

// ---- context-free forms (what a `NativeLoader`-style unit struct binds: loader.rs:108 has no &self) ----
static std::mutex g_default_mu;
static snarkv_ctx* g_default_ctx = nullptr;
static std::recursive_mutex g_default_call_mu;  // one shared context: calls from different host threads take turns
static int default_ctx(snarkv_ctx** out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (!g_default_ctx) {
    int rc = snarkv_pallas_ctx_create(0, nullptr, &g_default_ctx);
    if (rc < 0) return rc;
  }
  *out = g_default_ctx;
  return SNARKV_OK;
}
#define PALLAS_DEFAULT_CTX()                                             \
  std::lock_guard<std::recursive_mutex> _call_lock(g_default_call_mu);   \
  snarkv_ctx* c;                                                         \
  SNARKV_TRY(default_ctx(&c))

int pallas_g1_msm_naive(const uint8_t* scalars32, const uint8_t* points64, size_t n, uint8_t out64[64]) {
  PALLAS_DEFAULT_CTX();
  return snarkv_pallas_g1_msm_naive(c, scalars32, points64, n, 0, out64);
}
int pallas_g1_msm_batched(const uint8_t* scalars32, const uint8_t* points64, const uint32_t* offsets, size_t n_msm,
                          uint8_t* out) {
  PALLAS_DEFAULT_CTX();
  return snarkv_pallas_g1_msm_batched(c, scalars32, points64, offsets, n_msm, 0, out);
}
int pallas_host_buffer(int slot, size_t bytes, void** out) {
  PALLAS_DEFAULT_CTX();
  return snarkv_pallas_ctx_host_buffer(c, slot, bytes, out);
}
int pallas_g1_msm_pippenger(const uint8_t* scalars32, const uint8_t* points64, size_t n, uint8_t out64[64]) {
  PALLAS_DEFAULT_CTX();
  return snarkv_pallas_g1_msm_pippenger(c, scalars32, points64, n, out64);
}
int pallas_ipa_dk_create(const uint8_t* g_points64, size_t n, snarkv_ipa_dk** out) {
  PALLAS_DEFAULT_CTX();
  return snarkv_pallas_ipa_dk_create(c, g_points64, n, out);
}
int pallas_ipa_decide_batch(const snarkv_ipa_dk* dk, const uint8_t* xi32, const uint8_t* u64, size_t m, uint8_t* ok) {
  PALLAS_DEFAULT_CTX();
  return snarkv_pallas_ipa_decide_batch(c, dk, xi32, u64, m, ok);
}

}  // extern "C"
