// The pasta build of the curve-generic part of the library -> libsnarkv_pallas.so
// (include/snarkv_pallas.h).  The reference's IPA layer is generic over `C: CurveAffine`
// and its own tests run it on pallas (snark-verifier/src/pcs/ipa.rs:434-466,
// pcs/ipa/accumulation.rs:240-290), whose only device-worthy work is
// `util::msm::multi_scalar_multiplication` (msm.rs:308-343: `IpaProvingKey::commit`,
// pcs/ipa.rs:220-229, and `IpaAs::decide`, pcs/ipa/decider.rs:47-55).
//
// Same sources as the BN254 library -- fq29.cuh / fr29.cuh / g1_29.cuh / msm_pippenger.hip /
// ipa.hip -- compiled with
//   -DSNARKV_CURVE_PALLAS   pallas_consts.h: p, r, b = 5 (curve_consts.h)
//   -DSNARKV_GLV=0          one virtual point per point, 255-bit digit source, 16 windows of 16
//                           bits (the GLV split in glv.cuh is BN-shaped and stays with BN254)
//   -Dsnarkv=snarkv_pallas  the C++ namespace, so both libraries can live in one process
// There is no pairing, no decider and no transcript here.
#include <stdarg.h>
#include <string.h>
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

}  // extern "C"
