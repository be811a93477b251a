/* snarkv_pallas.h -- C ABI of the pasta build (libsnarkv_pallas.so): the large MSM and the IPA
 * decider on pallas, y^2 = x^3 + 5 over p = 2^254 + 45560315531419706090280762371685220353 with
 * group order q = 2^254 + 45560315531506369815346746415080538113 (halo2curves `pasta::pallas`).
 *
 * Why it exists: the reference's IPA layer is generic over `C: CurveAffine` and is TESTED on
 * pallas only (snark-verifier/src/pcs/ipa.rs:434-466, pcs/ipa/accumulation.rs:240-290,
 * system/halo2/test/ipa/native.rs); those tests are the reference's only consumers of
 * `util::msm::multi_scalar_multiplication` (msm.rs:308-343).
 *
 * Same conventions as snarkv_amd.h: scalars 32-byte little-endian canonical (< q), points
 * x || y 64 bytes little-endian canonical (< p), identity = 64 zero bytes; return 0 / negative
 * SNARKV_ERR_*; `snarkv_ctx` is this library's own context type (do not mix handles between the two
 * libraries).  The product path has no CPU fallback.                                          */
#ifndef SNARKV_PALLAS_H
#define SNARKV_PALLAS_H
#include "snarkv_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

int snarkv_pallas_ctx_create(int device, void* hip_stream, snarkv_ctx** out);
void snarkv_pallas_ctx_destroy(snarkv_ctx* ctx);
int snarkv_pallas_ctx_sync(snarkv_ctx* ctx);
/* stream ordering without a host round trip, as snarkv_ctx_wait_stream / snarkv_stream_wait_ctx / snarkv_ctx_stream */
int snarkv_pallas_ctx_wait_stream(snarkv_ctx* ctx, void* hip_stream);
int snarkv_pallas_stream_wait_ctx(snarkv_ctx* ctx, void* hip_stream);
void* snarkv_pallas_ctx_stream(snarkv_ctx* ctx);
int snarkv_pallas_ctx_host_buffer(snarkv_ctx* ctx, int slot, size_t bytes, void** out); /* as snarkv_ctx_host_buffer */
const char* snarkv_pallas_last_error(void);
const char* snarkv_pallas_version(void);

/* `util::msm::multi_scalar_multiplication(&[C::Scalar], &[C]) -> C::Curve` for C = pallas::Affine
 * (reference snark-verifier/src/util/msm.rs:308-343), normalised to affine.  n = 0 ->
 * SNARKV_ERR_EMPTY (reference: index panic, msm.rs:265).                                       */
int snarkv_pallas_g1_msm_pippenger(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                                   uint8_t out64[64]);
int snarkv_pallas_g1_msm_pippenger_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64, size_t n,
                                       int window_bits, void* d_out64);

/* `NativeLoader::multi_scalar_multiplication(&[(&Scalar, &C)]) -> C` for C = pallas::Affine (reference
 * snark-verifier/src/loader/native.rs:61-71; trait loader.rs:108-112) -- one MSM, or n_msm of them as
 * segments of one launch (segment k = terms offsets[k] .. offsets[k+1], offsets[0] = 0).  An empty
 * MSM -> SNARKV_ERR_EMPTY (reference: `reduce().unwrap()` panic, native.rs:69).  flags:
 * SNARKV_FLAG_VALIDATE checks canonical encodings and curve membership on the device.        */
int snarkv_pallas_g1_msm_naive(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                               uint32_t flags, uint8_t out64[64]);
int snarkv_pallas_g1_msm_batched(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64,
                                 const uint32_t* offsets, size_t n_msm, uint32_t flags, uint8_t* out);

/* `AccumulationDecider::{decide, decide_all}` for `IpaAs<pallas::Affine, _>` (reference
 * snark-verifier/src/pcs/ipa/decider.rs:47-66); semantics as snarkv_ipa_* in snarkv_amd.h.     */
int snarkv_pallas_ipa_dk_create(snarkv_ctx* ctx, const uint8_t* g_points64, size_t n, snarkv_ipa_dk** out);
void snarkv_pallas_ipa_dk_destroy(snarkv_ipa_dk* dk);
uint32_t snarkv_pallas_ipa_dk_k(const snarkv_ipa_dk* dk);
int snarkv_pallas_ipa_dk_create_shard(snarkv_ctx* ctx, const uint8_t* g_shard64, size_t count, uint32_t k, size_t first,
                                      snarkv_ipa_dk** out);
int snarkv_pallas_ipa_commit_partial_dev(snarkv_ctx* ctx, const snarkv_ipa_dk* dk, const uint8_t* xi32, void* d_partial);
int snarkv_pallas_ipa_decide_batch(snarkv_ctx* ctx, const snarkv_ipa_dk* dk, const uint8_t* xi32, const uint8_t* u64,
                                   size_t m, uint8_t* ok);

/* Context-free forms over a lazily created process-global context (device 0), as the bn254_* entry
 * points of snarkv_amd.h: `multi_scalar_multiplication` has no `&self` in the reference (loader.rs:108). */
int pallas_g1_msm_naive(const uint8_t* scalars32, const uint8_t* points64, size_t n, uint8_t out64[64]);
int pallas_g1_msm_batched(const uint8_t* scalars32, const uint8_t* points64, const uint32_t* offsets, size_t n_msm,
                          uint8_t* out);
int pallas_g1_msm_pippenger(const uint8_t* scalars32, const uint8_t* points64, size_t n, uint8_t out64[64]);
int pallas_host_buffer(int slot, size_t bytes, void** out); /* as bn254_host_buffer */
int pallas_ipa_dk_create(const uint8_t* g_points64, size_t n, snarkv_ipa_dk** out);
int pallas_ipa_decide_batch(const snarkv_ipa_dk* dk, const uint8_t* xi32, const uint8_t* u64, size_t m, uint8_t* ok);

#ifdef __cplusplus
}
#endif
#endif
