/* snarkv_host.h -- C ABI of the host mirror (libsnarkv_host.so): the reference's verifier-side API above the
 * loader, for callers that hand over BYTES (protocol, instances, proofs) and want accumulators / a verdict,
 * with every EC operation on the MI355X (through libsnarkv_amd.so, include/snarkv_amd.h).
 *
 * What each entry point stands for in the reference (snark-verifier/src/...):
 *   snarkv_host_protocol_parse            a `PlonkProtocol` value (verifier/plonk/protocol.rs:19-71); format 0 = the
 *                                         packed LE form of host/wire.hpp, format 1 = the reference's serde-JSON form
 *   snarkv_host_dk_create                 `KzgDecidingKey::new(g1, g2, s_g2)` (pcs/kzg/decider.rs:6-42); G2 line tables
 *                                         are built once per key on the device
 *   snarkv_host_plonk_succinct_verify_batch
 *                                         N x `PlonkSuccinctVerifier::{read_proof, verify}` (verifier/plonk.rs:58-92) with
 *                                         all 2N MSMs in ONE segmented device launch; output: every `KzgAccumulator`
 *   snarkv_host_kzg_as_accumulate         `KzgAs::create_proof` (non-zk; pcs/kzg/accumulation.rs:148-197) over m accumulators
 *                                         with a fresh Keccak (EVM) transcript, as examples/evm-verifier-with-accumulator.rs:357-380
 *   snarkv_host_kzg_as_create_proof / _verify  the same in full: zk blind branch (accumulation.rs:163-175), Keccak or
 *                                         Poseidon transcript, and `KzgAsProof::read` + `verify` on caller-supplied bytes
 *   snarkv_host_kzg_decide / _decide_all  `AccumulationDecider::{decide, decide_all}` (pcs/kzg/decider.rs:70-93)
 *   snarkv_host_aggregate                 the whole job: succinct-verify N proofs -> accumulate -> decide
 *   snarkv_host_aggregate_many            several such jobs sharing their three device launches
 *   snarkv_host_plonk_verify              `PlonkVerifier::verify` (verifier/plonk.rs:133: succinct verify + decide_all)
 *
 * Byte layouts: Fr / Fq 32 B little-endian canonical; G1 64 B x|y (identity = zeros); G2 128 B x.c0|x.c1|y.c0|y.c1;
 * accumulator 128 B lhs|rhs; protocol / instances / proofs as documented in snark-verifier_amd/host/wire.hpp
 * (instances: per proof `u32 columns, per column u32 m, m x Fr`; proofs: per proof `u32 len, bytes`).
 * Ownership: the caller owns every buffer; handles are freed with the matching *_free.
 * Return codes (the reference's `Result<_, Error>` / panics, lib.rs:18-28):
 *   1 accept / done, 0 reject (`Error::AssertionFailure`: the pairing check failed),
 *   SNARKV_HOST_ERR_* below for the other `Error` variants and for panics of the reference (unwrap / assert).
 * Thread safety: handles are immutable after creation and may be shared; calls may come from any thread (device
 * work is serialised on the process-global device context, as `NativeLoader` is a process-global unit struct).
 * There is NO CPU fallback: without libsnarkv_amd.so or a HIP device every device-touching call fails.
 */
#ifndef SNARKV_HOST_H
#define SNARKV_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNARKV_HOST_ERR_TRANSCRIPT (-10)        /* Error::Transcript (short / malformed proof bytes)           */
#define SNARKV_HOST_ERR_INVALID_INSTANCES (-11) /* Error::InvalidInstances                                      */
#define SNARKV_HOST_ERR_INVALID_PROTOCOL (-12)  /* Error::InvalidProtocol                                       */
#define SNARKV_HOST_ERR_OTHER (-13)
#define SNARKV_HOST_ERR_TRAILING (-14)          /* bytes left in a proof after read_proof (strict mode only)    */
#define SNARKV_HOST_ERR_CAPACITY (-6)           /* output buffer too small                                      */
#define SNARKV_HOST_ERR_ARG (-5)                /* null pointer / unknown enum value                            */
#define SNARKV_HOST_ERR_PANIC (-100)            /* the reference would panic here (unwrap / assert); see last_error */
#define SNARKV_HOST_ERR_DEVICE (-101)           /* device / runtime failure; see snarkv_host_last_error()       */

#define SNARKV_HOST_MOS_GWC19 0  /* pcs/kzg/multiopen/gwc19.rs  */
#define SNARKV_HOST_MOS_BDFG21 1 /* pcs/kzg/multiopen/bdfg21.rs */

#define SNARKV_HOST_TRANSCRIPT_EVM 0             /* system/halo2/transcript/evm.rs (Keccak-256)                          */
#define SNARKV_HOST_TRANSCRIPT_POSEIDON 1        /* system/halo2/transcript/halo2.rs (T=5, RATE=4, R_F=8, R_P=60), host  */
#define SNARKV_HOST_TRANSCRIPT_POSEIDON_DEVICE 2 /* the same transcript, hashed for the whole batch on the device       */
/* the same transcript, hashed wherever the batch is faster: on the host threads below SNARKV_HOST_POSEIDON_DEVICE_MIN
 * proofs (the device launch is one latency chain of ~3 ms whatever the batch: measured 2 x slower than 64 host threads at
 * 64 proofs, 1.4 x faster at 1 024), on the device from there on; three times that threshold on a CPU with AVX-512 IFMA,
 * whose sponge is four times faster (host/poseidon_ifma.hpp).  ONE job (snarkv_host_aggregate) on a host with 32 or more pool
 * threads stays on the host route at every size from SNARKV_HOST_PIPELINE_MIN proofs on: there the route is a pipeline
 * bounded by the accumulation sponge alone (see snarkv_host_aggregate).  Same bytes either way. */
#define SNARKV_HOST_TRANSCRIPT_POSEIDON_AUTO 3
#define SNARKV_HOST_POSEIDON_DEVICE_MIN 512

#define SNARKV_HOST_PROTOCOL_PACKED 0     /* host/wire.hpp                                                     */
#define SNARKV_HOST_PROTOCOL_SERDE_JSON 1 /* serde_json of `PlonkProtocol<G1Affine>` (protocol.rs:19-71)         */
#define SNARKV_HOST_PROTOCOL_BINCODE 2    /* bincode 1.x (default options) of the same struct                   */

typedef struct snarkv_host_protocol snarkv_host_protocol;
typedef struct snarkv_host_dk snarkv_host_dk;
typedef struct snarkv_host_snark snarkv_host_snark;

/* thread-local message of the last failing call on this thread */
const char* snarkv_host_last_error(void);

int snarkv_host_protocol_parse(const uint8_t* bytes, size_t len, int format, snarkv_host_protocol** out);
void snarkv_host_protocol_free(snarkv_host_protocol* p);
/* re-serialises to the packed form (format 0); returns the length, or SNARKV_HOST_ERR_CAPACITY (needed length in *len_out) */
int snarkv_host_protocol_pack(const snarkv_host_protocol* p, uint8_t* out, size_t cap, size_t* len_out);

/* The SDK's `Snark { protocol, instances, proof }` (snark-verifier-sdk/src/lib.rs:47-53): format 2 = the bincode file
 * `gen_snark` writes / `read_snark` reads (snark-verifier-sdk/src/halo2.rs:266-281,313-316), format 1 = its serde_json form.
 * The accessors hand the parts back in this library's wire forms (protocol handle borrowed from the snark; instances as
 * one packed instance block; proof bytes), ready for the verify entry points below. */
int snarkv_host_snark_parse(const uint8_t* bytes, size_t len, int format, snarkv_host_snark** out);
void snarkv_host_snark_free(snarkv_host_snark* s);
const snarkv_host_protocol* snarkv_host_snark_protocol(const snarkv_host_snark* s);
int snarkv_host_snark_instances(const snarkv_host_snark* s, uint8_t* out, size_t cap, size_t* len_out);
int snarkv_host_snark_proof(const snarkv_host_snark* s, uint8_t* out, size_t cap, size_t* len_out);

int snarkv_host_dk_create(const uint8_t g1[64], const uint8_t g2[128], const uint8_t s_g2[128], snarkv_host_dk** out);
void snarkv_host_dk_free(snarkv_host_dk* dk);

/* N proofs of one protocol -> their accumulators (the new one of each proof first, then the old ones it carried in its
 * instances), 128 B each, in proof order.  strict != 0 also rejects proofs with trailing bytes. */
int snarkv_host_plonk_succinct_verify_batch(const snarkv_host_protocol* protocol, const snarkv_host_dk* dk, int mos,
                                            int transcript, const uint8_t* instances, size_t instances_len,
                                            const uint8_t* proofs, size_t proofs_len, uint32_t n, int strict,
                                            uint8_t* accs_out, size_t accs_cap, uint32_t* n_accs);

/* m accumulators -> one (KzgAs::create_proof, non-zk, fresh EVM transcript).  r_out32 (optional): the challenge r. */
int snarkv_host_kzg_as_accumulate(const uint8_t* accs128, uint32_t m, uint8_t acc_out[128], uint8_t* r_out32);
/* `KzgAs::create_proof` in full (pcs/kzg/accumulation.rs:148-197) over a FRESH transcript of kind `transcript`
 * (SNARKV_HOST_TRANSCRIPT_EVM or _POSEIDON).  pk_g_sg128 = NULL: non-zk.  Otherwise the proving key's (g, s_g) pair
 * (2 x 64 B) and `blind_scalar32` (32 B LE canonical Fr: what the reference draws from its rng) give the zk branch
 * (accumulation.rs:163-175): the blind pair (s_g * b, g * b) is written to the transcript and accumulated last.
 * Out: the accumulator, the bytes the transcript produced (the `KzgAsProof` a verifier reads; empty for non-zk;
 * *proof_len is always set, SNARKV_HOST_ERR_CAPACITY when proof_cap is too small), and the challenge r. */
int snarkv_host_kzg_as_create_proof(const uint8_t* accs128, uint32_t m, int transcript, const uint8_t* pk_g_sg128,
                                    const uint8_t* blind_scalar32, uint8_t acc_out[128], uint8_t* proof_out,
                                    size_t proof_cap, size_t* proof_len, uint8_t* r_out32);
/* `KzgAsProof::read` + `KzgAs::verify` (accumulation.rs:114-137, 41-63) on CALLER-SUPPLIED proof bytes: the verifier's
 * side of the call above (zk != 0: the proof carries the blind pair).  Bytes left over -> SNARKV_HOST_ERR_TRAILING. */
int snarkv_host_kzg_as_verify(const uint8_t* accs128, uint32_t m, int transcript, int zk, const uint8_t* proof,
                              size_t proof_len, uint8_t acc_out[128], uint8_t* r_out32);

/* 1 accept / 0 reject.  decide_all: one batched device launch; ok_out (optional) = per-accumulator verdicts. */
int snarkv_host_kzg_decide(const snarkv_host_dk* dk, const uint8_t acc128[128]);
int snarkv_host_kzg_decide_all(const snarkv_host_dk* dk, const uint8_t* accs128, uint32_t m, uint8_t* ok_out);

/* succinct-verify N proofs, accumulate, decide.  host_threads: threads for the transcript / Fr front half (0 = all).
 * timings_ms (optional, 6 doubles): read_proofs, fr_algebra, msm_device, accumulate, decide, total.
 * acc_out (optional): the aggregated accumulator (also written on reject).
 * Poseidon proofs hashed on the host, SNARKV_HOST_PIPELINE_MIN (environment; default 256; 0 = never) proofs or more: the
 * job runs as a pipeline -- proofs read on the host pool, their MSMs launched chunk by chunk, the accumulation transcript
 * (one sponge over 4 n field elements: n + 1 dependent permutations on one thread) absorbing under both.  Same accumulator,
 * same verdict, same error for a bad batch (the first bad proof in proof order) as the unpipelined job; read_proofs,
 * fr_algebra and msm_device are then the helper threads' busy times (they run under `accumulate`), total is wall time. */
int snarkv_host_aggregate(const snarkv_host_protocol* protocol, const snarkv_host_dk* dk, int mos, int transcript,
                          const uint8_t* instances, size_t instances_len, const uint8_t* proofs, size_t proofs_len,
                          uint32_t n, unsigned host_threads, double* timings_ms, uint8_t* acc_out);

/* SEVERAL such jobs in one call (a service batching its requests): `job_sizes[k]` consecutive proofs of the two blobs
 * belong to job k.  Per job exactly what snarkv_host_aggregate returns for its proofs -- accs_out: n_jobs x 128 bytes,
 * ok_out: the pairing verdict per job -- but the device runs three launches whatever n_jobs is (every proof's MSMs,
 * every job's two KzgAs MSMs, every job's pairing check): the launches of a small job are latency chains that fill a
 * fraction of the GPU, and 16 jobs of 64 proofs cost what one job of 1 024 does.  Returns 1 if every job is accepted,
 * 0 if some job is rejected (see ok_out), a negative code for malformed input (no partial results).              */
int snarkv_host_aggregate_many(const snarkv_host_protocol* protocol, const snarkv_host_dk* dk, int mos, int transcript,
                               const uint8_t* instances, size_t instances_len, const uint8_t* proofs, size_t proofs_len,
                               const uint32_t* job_sizes, uint32_t n_jobs, unsigned host_threads, double* timings_ms,
                               uint8_t* accs_out, uint8_t* ok_out);

/* PlonkVerifier::verify on N proofs: succinct verify (one launch) then ONE decide_all over every accumulator. */
int snarkv_host_plonk_verify(const snarkv_host_protocol* protocol, const snarkv_host_dk* dk, int mos, int transcript,
                             const uint8_t* instances, size_t instances_len, const uint8_t* proofs, size_t proofs_len,
                             uint32_t n);

/* `LimbsEncoding<LIMBS = 4, BITS = 68>` (pcs/kzg/accumulator.rs:34-82, util/arithmetic.rs:270-298): accumulator <-> 16 Fr */
int snarkv_host_accumulator_to_limbs(const uint8_t acc128[128], uint8_t limbs_out[16 * 32]);
int snarkv_host_accumulator_from_limbs(const uint8_t limbs[16 * 32], uint8_t acc_out[128]);

#ifdef __cplusplus
}
#endif
#endif
