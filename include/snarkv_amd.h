/* snarkv_amd.h -- C ABI of the MI355X-native KZG-accumulation hot path.
 *
 * This is the drop-in boundary SURVEY.md section 8(b) defines: the entry points
 * a Rust `impl EcPointLoader / AccumulationDecider for GpuNativeLoader` would
 * bind through `extern "C"` (binding stub: INTEGRATION.md).  Plain pointers and
 * sizes only; caller owns every buffer; nothing allocated here crosses back.
 *
 * Byte layouts (what the reference serialises):
 *   Fr, Fq    32-byte little-endian CANONICAL integer  (= `PrimeField::to_repr`,
 *             reference snark-verifier/src/util/msm.rs:264)
 *   G1Affine  x || y, 64 bytes; identity = 64 zero bytes (halo2curves (0,0))
 *   Fq2       c0 || c1, 64 bytes;  G2Affine  x || y, 128 bytes
 *   KzgAccumulator  lhs || rhs, 128 bytes
 *             (reference snark-verifier/src/pcs/kzg/accumulator.rs:6-26)
 *
 * Scalars and coordinates MUST be canonical (< r resp. < p) and points on the
 * curve -- the reference's types guarantee this at its boundary; pass
 * SNARKV_FLAG_VALIDATE to have it checked (returns SNARKV_ERR_ENCODING).
 *
 * Return convention: 0 = SNARKV_OK, negative = error.  `decide` returns
 * 1 accept / 0 reject / negative error  (reference: Ok(()) /
 * Err(Error::AssertionFailure), snark-verifier/src/pcs/kzg/decider.rs:78-81).
 * The reference PANICS on an empty MSM (native.rs:69 `reduce().unwrap()`,
 * msm.rs:265 `scalars[0]`) and on a length mismatch (msm.rs:309); here those
 * are SNARKV_ERR_EMPTY / SNARKV_ERR_LENGTH.
 *
 * Threading: a context owns one HIP stream and its scratch; calls on one
 * context are serialised by the caller (one context per host thread).  The
 * context-free `bn254_*` entry points use a lazily created process-global
 * context on device 0 / HIP_VISIBLE_DEVICES, matching the reference's static
 * dispatch (`EcPointLoader::multi_scalar_multiplication` has no `&self`,
 * snark-verifier/src/loader.rs:108); they are thread-safe (calls from
 * different host threads take turns on that one context -- the reference's
 * NativeLoader is `Sync`).
 */
#ifndef SNARKV_AMD_H
#define SNARKV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNARKV_OK 0
#define SNARKV_ERR_EMPTY (-1)    /* n == 0 or an empty segment           */
#define SNARKV_ERR_LENGTH (-2)   /* inconsistent sizes / offsets          */
#define SNARKV_ERR_ENCODING (-3) /* non-canonical field element, off-curve */
#define SNARKV_ERR_DEVICE (-4)   /* HIP error; see snarkv_last_error()    */
#define SNARKV_ERR_ARG (-5)      /* null pointer / bad handle             */

#define SNARKV_FLAG_VALIDATE 1u
/* Field elements in halo2curves' IN-MEMORY form instead of the wire form: Fr / Fq as the four little-endian u64 limbs of
 * a * 2^256 mod r (mod p) -- what `Fr`, `Fq`, `G1Affine { x, y }`, `G2Affine` hold in memory (`[u64; 4]`, Montgomery
 * R = 2^256; the identity is (0, 0) in both forms), so a Rust caller passes `&[Fr]` / `&[G1Affine]` as they lie and gets
 * a `G1Affine` back, with no `to_repr()` / `from_repr()` per element (reference util/msm.rs:308 takes exactly those slices;
 * msm.rs:264 converts the scalars itself).  Covers scalars, G1 / G2 points and accumulators in, G1 points out of every
 * BN254 entry point below except the IPA, Poseidon-transcript and test-hook ones (`snarkv_kzg_pairing_value` returns
 * canonical Gt bytes; compressed points into `snarkv_g1_decompress` are the wire form by definition, its output follows
 * the flag).  Cost on the device: the points enter the kernels' 9 x 29-bit domain by one product either way (another
 * constant); a scalar costs one Fr product more (k_prepare +12 %, ~1 % of a 2^20-point MSM). */
#define SNARKV_FLAG_MONTGOMERY 2u

typedef struct snarkv_ctx snarkv_ctx;
typedef struct snarkv_dk snarkv_dk;
typedef struct snarkv_poseidon snarkv_poseidon;

/* ---- context ---------------------------------------------------------- */
/* `hip_stream` may be NULL (the context creates its own stream) or an
 * existing hipStream_t the caller wants the work enqueued on (e.g. the
 * current torch stream).                                                   */
int snarkv_ctx_create(int device, void* hip_stream, snarkv_ctx** out);
void snarkv_ctx_destroy(snarkv_ctx* ctx);
int snarkv_ctx_sync(snarkv_ctx* ctx);
/* Stream ordering WITHOUT a host round trip (SURVEY.md 8b "Threading": the `_dev` entry points are asynchronous on the
 * context's stream; a caller that fills the inputs or reads the outputs on ANOTHER stream orders the two here):
 *   snarkv_ctx_wait_stream   everything enqueued on `hip_stream` so far happens-before whatever the context enqueues next
 *                            (inputs written by the caller's stream: call it BEFORE the `_dev` entry point);
 *   snarkv_stream_wait_ctx   everything the context has enqueued so far happens-before whatever `hip_stream` runs next
 *                            (outputs read by the caller's stream / a collective on it: call it AFTER the entry point).
 * `hip_stream` = a hipStream_t OF THE CONTEXT'S DEVICE; NULL is the legacy default stream.  Both are an event record + hipStreamWaitEvent: they
 * return at once and cost no host synchronisation.  A context that was created ON `hip_stream` is ordered already: no-op.
 * `snarkv_ctx_stream` returns the hipStream_t the context enqueues on (for callers that bring their own events).     */
int snarkv_ctx_wait_stream(snarkv_ctx* ctx, void* hip_stream);
int snarkv_stream_wait_ctx(snarkv_ctx* ctx, void* hip_stream);
void* snarkv_ctx_stream(snarkv_ctx* ctx);
/* Pinned host memory owned by the context, for callers that assemble their inputs themselves: the host-pointer
 * entry points (`snarkv_g1_msm_batched`, `snarkv_g1_msm_pippenger`, ...) copy with hipMemcpyAsync, which is a DMA
 * out of pinned memory and a bounce copy out of pageable memory (2.4 MB of MSM terms: ~0.1 ms against ~0.3 ms).
 * SNARKV_HOST_BUFFERS independent slots, grow-only; a pointer stays valid until the same slot is requested with a
 * larger size (or the context is destroyed).  One user per context at a time, like every call on a context.   */
#define SNARKV_HOST_BUFFERS 4
int snarkv_ctx_host_buffer(snarkv_ctx* ctx, int slot, size_t bytes, void** out);
const char* snarkv_last_error(void);
const char* snarkv_version(void);
/* Default flags of a context (SNARKV_FLAG_*): OR-ed into the `flags` argument of every call on it, and THE flags of the
 * entry points that have no such argument (`*_dev`, `*_many_*`, the sampling utilities, `snarkv_g1_decompress`).  A
 * `snarkv_mgpu` handle's ranks are contexts of their own (`snarkv_mgpu_ctx`); `bn254_set_flags` /
 * `bn254_set_thread_flags` set those of the context-free calls.  Unknown bits are SNARKV_ERR_ARG. */
int snarkv_ctx_set_flags(snarkv_ctx* ctx, uint32_t flags);
uint32_t snarkv_ctx_get_flags(const snarkv_ctx* ctx);

/* ---- A3: NativeLoader::multi_scalar_multiplication --------------------- *
 * replaces snark-verifier/src/loader/native.rs:61-71
 *   pairs.iter().map(|(s, b)| *b * s).reduce(|a, v| a + v).unwrap().to_affine()
 * host buffers in, 64-byte affine point out.                               */
int snarkv_g1_msm_naive(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                        uint32_t flags, uint8_t out64[64]);

/* Segmented form: n_msm independent MSMs in ONE launch.  MSM k owns terms
 * [offsets[k], offsets[k+1]); offsets has n_msm+1 entries, offsets[0] = 0.
 * This is the shape of the accumulation path: per proof the two `evaluate`
 * calls of Gwc19/Bdfg21::verify (gwc19.rs:79-80, bdfg21.rs:80-81) and the two
 * of KzgAs::verify (accumulation.rs:53-60).  out = n_msm * 64 bytes.        */
int snarkv_g1_msm_batched(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64,
                          const uint32_t* offsets, size_t n_msm, uint32_t flags, uint8_t* out);

/* ---- A1: util::msm::multi_scalar_multiplication ------------------------ *
 * replaces snark-verifier/src/util/msm.rs:308-343 (windowed-bucket
 * Pippenger).  The reference returns a projective `C::Curve`; this returns
 * its `to_affine()` (the only representation-independent form).            */
int snarkv_g1_msm_pippenger(snarkv_ctx* ctx, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                            uint32_t flags, uint8_t out64[64]);

/* Device-resident variants (inputs and output already in HBM; asynchronous
 * on the context stream -- call snarkv_ctx_sync or sync the stream).  These
 * are what bench.py times.  `window_bits` = 0 picks the default.           */
int snarkv_g1_msm_pippenger_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64, size_t n,
                                int window_bits, void* d_out64);
int snarkv_g1_msm_batched_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64,
                              const void* d_offsets, size_t n_msm, size_t n_terms, void* d_out);

/* MANY independent large MSMs in one call (what a batch of `util::msm::multi_scalar_multiplication` calls,
 * msm.rs:308, is to the reference: the accumulation path issues one per `Msm::evaluate`): d_out64s[64 i ..] =
 * sum_j scalars_i[j] * points_i[j], the same bytes `snarkv_g1_msm_pippenger_dev` gives for each.  One call, one
 * context: the library pipelines the MSMs itself (sorts on high-priority streams, accumulations back to back, ONE
 * batched tail per round; csrc/capi.hip) -- 4-5 % faster than four single calls kept in flight for 8-20 MSMs of 2^20
 * points, level beyond.  Scratch: ~0.5 KiB per point and job; more than SNARKV_MANY_MAX_JOBS MSMs (or 48 GiB of
 * scratch) run as successive rounds; an MSM large enough for the chunk pipeline (3 * 2^20 points) makes the call
 * fall back to one MSM after the other.  Asynchronous on the context stream; n[i] = 0 is SNARKV_ERR_EMPTY. */
#define SNARKV_MANY_MAX_JOBS 64
int snarkv_g1_msm_pippenger_many_dev(snarkv_ctx* ctx, size_t count, const void* const* d_scalars32,
                                     const void* const* d_points64, const size_t* n, int window_bits, void* d_out64s);
/* The batch with HOST-resident inputs (what a caller holding `&[Fr]` / `&[G1Affine]` slices has; with
 * SNARKV_FLAG_MONTGOMERY they are passed as they lie in memory): job i's 96 n[i] bytes are uploaded on a copy stream of the
 * context, and its kernels wait for ITS upload only, so the link and the GPU work at the same time -- a batch of 2^20-point
 * MSMs runs at the PCIe rate (~1.8 ms per MSM on Gen5 x16) instead of upload + 1.5 ms each.  Pin the sources
 * (snarkv_ctx_host_buffer, or snarkv_host_register on memory the caller owns): pageable memory is copied through the
 * runtime's bounce buffer at about a third of the rate.  Synchronous: out64s[64 i ..] is valid on return. */
int snarkv_g1_msm_pippenger_many(snarkv_ctx* ctx, size_t count, const uint8_t* const* scalars32,
                                 const uint8_t* const* points64, const size_t* n, uint32_t flags, uint8_t* out64s);
/* hipHostRegister / hipHostUnregister for callers without a HIP binding (pin a Vec's buffer once, reuse it for many calls) */
int snarkv_host_register(void* p, size_t bytes);
int snarkv_host_unregister(void* p);
/* the same with projective partials out (count x SNARKV_G1_PARTIAL_BYTES): a rank's shard of `count` multi-GPU MSMs --
 * ONE all-gather of all the partials, then snarkv_g1_fold_partials_dev per MSM                                        */
int snarkv_g1_msm_pippenger_many_partial_dev(snarkv_ctx* ctx, size_t count, const void* const* d_scalars32,
                                             const void* const* d_points64, const size_t* n, int window_bits,
                                             void* d_partials);

/* Multi-GPU building blocks: each rank reduces ITS shard of
 * the points to one projective partial (SNARKV_G1_PARTIAL_BYTES, internal XYZZ
 * 9x29-bit Montgomery form, opaque), the partials are all-gathered (RCCL), and every rank folds
 * them to the same affine result.                                           */
#define SNARKV_G1_PARTIAL_BYTES 144
int snarkv_g1_msm_pippenger_partial_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64,
                                        size_t n, int window_bits, void* d_partial);
int snarkv_g1_fold_partials_dev(snarkv_ctx* ctx, const void* d_partials, size_t count, void* d_out64);
/* `jobs` folds in one launch (the multi-GPU form of the batch entry point): d_partials = [job][count] partials --
 * the all-gathered [rank][job] array transposed --, d_out64s = [job] affine points                                   */
int snarkv_g1_fold_partials_many_dev(snarkv_ctx* ctx, const void* d_partials, size_t count, size_t jobs, void* d_out64s);

/* ---- A8: KzgAs::decide / decide_all ------------------------------------ *
 * replaces snark-verifier/src/pcs/kzg/decider.rs:70-93.  The deciding key
 * (KzgDecidingKey{svk.g, g2, s_g2}, decider.rs:6-42) is loaded once; its two
 * G2 line tables (the reference's per-call `G2Prepared::from`, decider.rs:74)
 * are computed on the device at load time.                                  */
int snarkv_dk_create(snarkv_ctx* ctx, const uint8_t g1_64[64], const uint8_t g2_128[128],
                     const uint8_t s_g2_128[128], uint32_t flags, snarkv_dk** out);
void snarkv_dk_destroy(snarkv_dk* dk);
/* 1 = accept, 0 = reject (reference: Err(AssertionFailure)), <0 = error */
int snarkv_kzg_decide(snarkv_ctx* ctx, const snarkv_dk* dk, const uint8_t acc128[128], uint32_t flags);
/* decide_all: ok[i] in {0,1} per accumulator; returns 1 iff all accept.     */
int snarkv_kzg_decide_batch(snarkv_ctx* ctx, const snarkv_dk* dk, const uint8_t* accs128, size_t m,
                            uint32_t flags, uint8_t* ok);
int snarkv_kzg_decide_batch_dev(snarkv_ctx* ctx, const snarkv_dk* dk, const void* d_accs128, size_t m,
                                void* d_ok);
/* Test hook: the fully exponentiated Gt value (12 x 32 bytes, tower order
 * c0.c0.c0 .. c1.c2.c1) of e(lhs,g2)*e(rhs,-s_g2).                          */
int snarkv_kzg_pairing_value(snarkv_ctx* ctx, const snarkv_dk* dk, const uint8_t acc128[128], uint8_t gt384[384]);

/* Canonical-coordinate + on-curve check of n G1 points (the `C::from_xy(..)`
 * of snark-verifier/src/pcs/kzg/accumulator.rs:76-77): SNARKV_OK or
 * SNARKV_ERR_ENCODING.                                                      */
int snarkv_g1_validate(snarkv_ctx* ctx, const uint8_t* points64, size_t n);

/* ---- compressed points of a Poseidon transcript ----------------------------- *
 * `C::from_bytes(&data)` of `PoseidonTranscript::read_ec_point`
 * (snark-verifier/src/system/halo2/transcript/halo2.rs:260-273) for a whole batch: n x 32 bytes (x little-endian,
 * bit 254 = parity of y, bit 255 = identity) -> n x 64 bytes x || y (the identity: zeros) and ok[i] = 1, or ok[i] = 0
 * where `from_bytes` is `None` (x >= p, no such point, malformed identity).  One square root per point: the host
 * front half of a 1 024-proof aggregation spent 2.4 ms there.  SNARKV_OK, or an error code for bad arguments.   */
int snarkv_g1_decompress(snarkv_ctx* ctx, const uint8_t* in32, size_t n, uint8_t* out64, uint8_t* ok);

/* ---- context-free entry points ------------------------------------------ *
 * `EcPointLoader::multi_scalar_multiplication` has no `&self` (loader.rs:108; the native loader is a unit struct,
 * native.rs:11-19), so a trait-bound caller reaches the device without a handle.  These forms draw a context (stream +
 * scratch) from a process-wide POOL per call: SNARKV_DEFAULT_CONTEXTS of them, else GPU_MAX_HW_QUEUES, else 4, created
 * on demand on device 0.  Calls from different host threads therefore run CONCURRENTLY on different contexts (16
 * rayon workers in flight = the `aggregate_*_pipelined` figures of bench.py); a thread returns to the context it used
 * last when that is free; with every context busy a call waits its turn.  All of them are synchronous (results are on
 * the host on return) and thread-safe. */
/* process default flags of the context-free calls, e.g. SNARKV_FLAG_MONTGOMERY once at start-up: every bn254_* call then
 * speaks halo2curves' in-memory form.  Takes effect for calls that START after it returns. */
int bn254_set_flags(uint32_t flags);
/* ... and an override for the CALLING THREAD only: flags >= 0 replace the process default for this thread's bn254_*
 * calls, -1 removes the override.  Returns the previous value (>= -1; -1 = none), or SNARKV_ERR_ARG (-5) when `flags`
 * has unknown bits or is below -1 -- the override is then UNCHANGED, and -5 must not be passed back as "the previous
 * value" (restore only what was returned with a value >= -1).  What a library that shares the process
 * with other bn254_* users needs (libsnarkv_host.so passes wire-form bytes whatever the application chose above). */
int64_t bn254_set_thread_flags(int64_t flags);
/* Releases what the context-free calls keep for the life of the process: the pool's contexts (streams + scratch), the
 * cached deciding-key line tables, the pinned buffers (bn254_host_buffer) of the calling thread and of threads that have
 * ended -- pointers handed out by bn254_host_buffer to THIS thread are invalid afterwards; buffers of other live threads
 * stay theirs.  SNARKV_ERR_ARG while a context-free call is in flight.  Later bn254_* calls start over lazily. */
int bn254_shutdown(void);
uint32_t bn254_get_flags(void); /* what the calling thread's next bn254_* call will use */
int bn254_default_contexts(int* created, int* cap); /* pool status: contexts created so far / the limit */
int bn254_g1_msm_naive(const uint8_t* scalars32, const uint8_t* points64, size_t n, uint8_t out64[64]);
int bn254_g1_msm_batched(const uint8_t* scalars32, const uint8_t* points64, const uint32_t* offsets, size_t n_msm,
                         uint8_t* out);
int bn254_g1_msm_pippenger(const uint8_t* scalars32, const uint8_t* points64, size_t n, uint8_t out64[64]);
int bn254_g1_decompress(const uint8_t* in32, size_t n, uint8_t* out64, uint8_t* ok);
/* pinned host memory for the inputs of the context-free calls, owned by the CALLING THREAD (SNARKV_HOST_BUFFERS slots,
 * grow-only; a pointer stays valid until the same thread asks for the same slot with a larger size): threads pack
 * side by side without a lock, whatever pool context their calls land on */
int bn254_host_buffer(int slot, size_t bytes, void** out);
int bn254_kzg_decide(const uint8_t g1_64[64], const uint8_t g2_128[128], const uint8_t s_g2_128[128],
                     const uint8_t acc128[128]);
int bn254_kzg_decide_batch(const uint8_t g1_64[64], const uint8_t g2_128[128], const uint8_t s_g2_128[128],
                           const uint8_t* accs128, size_t m, uint8_t* ok);
/* deciding key kept across calls (its G2 line tables stay in HBM) */
int bn254_kzg_dk_create(const uint8_t g1_64[64], const uint8_t g2_128[128], const uint8_t s_g2_128[128],
                        snarkv_dk** out);
int bn254_kzg_dk_decide_batch(const snarkv_dk* dk, const uint8_t* accs128, size_t m, uint8_t* ok);
int bn254_g1_validate(const uint8_t* points64, size_t n);

/* ---- synthetic inputs generated in HBM (bench/test utility) --- *
 * Seeded SplitMix64 streams -> canonical Fr scalars / G1 points, element
 * indices [first, first+n).  Not a reference function: the reference draws
 * its test inputs from OsRng / ChaCha20 (system/halo2/test.rs:191).          */
int snarkv_sample_scalars_dev(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_scalars32);
int snarkv_sample_points_dev(snarkv_ctx* ctx, uint64_t seed, uint64_t first, size_t n, void* d_points64);

/* ---- profiling hooks used by bench.py --------------------------------- *
 * Per-stage HIP-event timings (ms) of the last *_dev Pippenger call on this
 * context when enabled: [0]=total [1]=prepare (GLV split, phi(P), Montgomery)
 * + digit histogram [2]=scan [3]=partition + level-2 sort [4]=bucket
 * accumulate [5]=bucket combine [6]=bucket reduce [7]=window sums + 2^(cw)
 * shift chains [8]=final sum + to_affine.                                    */
#define SNARKV_PIP_STAGES 9
/* Integer-VALU roofline probe (bench.py): peak rate, with 4 wavefronts per SIMD
 * on every CU, of which = 0: dependent 254-bit Montgomery products (Fq-mul/s),
 * which = 1: whole mixed G1 additions (madd/s) -- the multiplier / adder of the
 * bucket-accumulate kernel in isolation (no memory traffic).                  */
int snarkv_ubench_valu(snarkv_ctx* ctx, int which, int iters, double* ops_per_s);
/* ---- bucket-sharded MSM, the "bucket-sum allreduce" of config 4 ----
 * Alternative to point-sharding + fold: every GPU accumulates the GLOBAL bucket grid
 * (windows x 2^(c-1) signed-digit buckets, c agreed through `..._bucket_geometry`
 * for the TOTAL n) from its shard of the points -- the `buckets[d-1].add_assign(base)`
 * half of util/msm.rs:291-296 -- the grids are exchanged by window range and added,
 * and each GPU runs the running-sum + `double()` half (msm.rs:285-302) on the
 * windows it owns; the projective partials are folded as in the point-sharded path.
 * A bucket is SNARKV_G1_PARTIAL_BYTES (144) bytes, the identity all-zero.            */
int snarkv_g1_msm_bucket_geometry(size_t n_total, int window_bits, uint32_t* c, uint32_t* windows,
                                  uint32_t* buckets_per_window);
int snarkv_g1_msm_fill_buckets_dev(snarkv_ctx* ctx, const void* d_scalars32, const void* d_points64, size_t n,
                                   int window_bits, void* d_buckets);
int snarkv_g1_buckets_add_dev(snarkv_ctx* ctx, void* d_dst, const void* d_src, size_t count);
int snarkv_g1_buckets_reduce_dev(snarkv_ctx* ctx, const void* d_buckets, uint32_t c, uint32_t w0, uint32_t wcount,
                                 void* d_partial);

/* ---- IPA decider: `AccumulationDecider::{decide, decide_all}` for `IpaAs` --------------------
 * reference snark-verifier/src/pcs/ipa/decider.rs:47-66:
 *     U == multi_scalar_multiplication(h_coeffs(xi, 1), dk.g).to_affine()
 * `snarkv_ipa_dk_create` uploads the committing key G (n = 2^k points, 64 B each: the `g` of
 * `IpaDecidingKey`, decider.rs:5-9) once; `snarkv_ipa_decide_batch` takes m accumulators
 * (`IpaAccumulator`, accumulator.rs:5-14: xi = k scalars of 32 B each, u = 64 B), builds h_coeffs
 * (pcs/ipa.rs:405-421) on the device, runs one 2^k-term Pippenger per accumulator and writes
 * ok[a] = 1 / 0 (0 is `Err(AssertionFailure("U == commit(G, h)"))`, not an error code).        */
typedef struct snarkv_ipa_dk snarkv_ipa_dk;
int snarkv_ipa_dk_create(snarkv_ctx* ctx, const uint8_t* g_points64, size_t n, snarkv_ipa_dk** out);
void snarkv_ipa_dk_destroy(snarkv_ipa_dk* dk);
uint32_t snarkv_ipa_dk_k(const snarkv_ipa_dk* dk);
int snarkv_ipa_decide_batch(snarkv_ctx* ctx, const snarkv_ipa_dk* dk, const uint8_t* xi32, const uint8_t* u64, size_t m,
                            uint8_t* ok);
/* Multi-GPU decide (the MSM is linear in the points, as util/msm.rs:322-335 chunks it): rank r keeps points
 * [first, first + count) of the key (`..._create_shard`), `snarkv_ipa_commit_partial_dev` leaves its share of
 * commit(G, h(xi)) as a SNARKV_G1_PARTIAL_BYTES projective partial in device memory; all-gather the partials,
 * fold them with snarkv_g1_fold_partials_dev and compare with U.  A shard key is refused by ..._decide_batch. */
int snarkv_ipa_dk_create_shard(snarkv_ctx* ctx, const uint8_t* g_shard64, size_t count, uint32_t k, size_t first,
                               snarkv_ipa_dk** out);
int snarkv_ipa_commit_partial_dev(snarkv_ctx* ctx, const snarkv_ipa_dk* dk, const uint8_t* xi32, void* d_partial);
int bn254_ipa_dk_create(const uint8_t* g_points64, size_t n, snarkv_ipa_dk** out);
int bn254_ipa_decide_batch(const snarkv_ipa_dk* dk, const uint8_t* xi32, const uint8_t* u64, size_t m, uint8_t* ok);

/* ---- Poseidon transcripts, batched (SURVEY.md 8f row N2 on the device) ----
 * Replaces, for MANY proofs at once, the hashing of the reference's native
 * `PoseidonTranscript` (snark-verifier/src/system/halo2/transcript/halo2.rs:170-321
 * over `Poseidon::{update, squeeze}`, snark-verifier/src/util/hash/poseidon.rs:145-202).
 * `snarkv_poseidon_create` takes the tables of the optimised schedule exactly as the
 * reference's `poseidon::Spec` holds them (poseidon.rs:166-201 reads
 * `spec.constants().{start, partial, end}` and
 * `spec.mds_matrices().{mds, pre_sparse_mds, sparse_matrices}`): every value a
 * 32-byte little-endian canonical Fr; matrices row-major; `start` has r_f/2 + 1
 * rows of t, `end` r_f/2 - 1 rows, one `row` (t) and one `col_hat` (t - 1) per
 * partial round.  2 <= t <= 8, rate < t.                                        */
int snarkv_poseidon_create(snarkv_ctx* ctx, uint32_t t, uint32_t rate, uint32_t r_f, uint32_t r_p,
                           const uint8_t* start, const uint8_t* partial, const uint8_t* end, const uint8_t* mds,
                           const uint8_t* pre_sparse_mds, const uint8_t* sparse_rows, const uint8_t* sparse_col_hats,
                           snarkv_poseidon** out);
void snarkv_poseidon_destroy(snarkv_poseidon* ps);
/* n transcripts of one shape: transcript i absorbs its L elements (elems + 32*L*i,
 * canonical Fr each) in S segments of seg_len[s] elements and squeezes after each
 * segment (sum of seg_len == L; a segment may be empty).  State starts at
 * `poseidon::State::default()` = [2^64, 0, ..].  out: n x S challenges, 32-byte LE. */
int snarkv_poseidon_transcript_batch(snarkv_ctx* ctx, const snarkv_poseidon* ps, const uint8_t* elems, size_t n,
                                     size_t L, const uint32_t* seg_len, size_t S, uint8_t* out);
int snarkv_poseidon_transcript_batch_dev(snarkv_ctx* ctx, const snarkv_poseidon* ps, const void* d_elems, size_t n,
                                         size_t L, const void* d_seg_len, size_t S, void* d_out);
/* The same hashing for n PROOFS of one protocol, read where they are: what a native PoseidonTranscript absorbs while it
 * reads a proof (system/halo2/transcript/halo2.rs:215-275) is, in the protocol's order, a value the caller brought
 * (initial state, instances), a scalar of the proof, or a coordinate of one of its compressed points reduced mod r
 * (`fe_to_fe`).  layout[k] (k < L) names the source of absorbed element k:  kind << 28 | value  with kind 0 = lead element
 * `value` of this proof (lead + 32 (n_lead i + value), canonical Fr), 1 = the 32-byte scalar at byte `value` of the proof,
 * 2 / 3 = x / y of point `value`, the compressed point (halo2curves bn256 `G1Affine::from_bytes`) at byte
 * point_offsets[value].  Proof i lies at proofs + stride i (stride and point offsets multiples of 16, scalar offsets of 4).
 * One pipeline on the device: every point decompressed, every transcript's input assembled, all transcripts hashed.
 * Out: n x S challenges; the n x P decompressed points (64 bytes, canonical whatever the context's flags) and a validity
 * byte each -- a proof with an invalid point or a non-canonical scalar gets meaningless challenges: the caller, who parses
 * the proof with them, rejects it there (host/aggregation.hpp does).                                                    */
int snarkv_poseidon_read_batch(snarkv_ctx* ctx, const snarkv_poseidon* ps, const uint8_t* proofs, size_t n, size_t stride,
                               const uint8_t* lead, size_t n_lead, const uint32_t* layout, size_t L,
                               const uint32_t* point_offsets, size_t P, const uint32_t* seg_len, size_t S,
                               uint8_t* challenges, uint8_t* points64, uint8_t* ok);

/* context-free forms (process-global context, thread-safe) */
int bn254_poseidon_create(uint32_t t, uint32_t rate, uint32_t r_f, uint32_t r_p, const uint8_t* start,
                          const uint8_t* partial, const uint8_t* end, const uint8_t* mds, const uint8_t* pre_sparse_mds,
                          const uint8_t* sparse_rows, const uint8_t* sparse_col_hats, snarkv_poseidon** out);
int bn254_poseidon_transcript_batch(const snarkv_poseidon* ps, const uint8_t* elems, size_t n, size_t L,
                                    const uint32_t* seg_len, size_t S, uint8_t* out);
int bn254_poseidon_read_batch(const snarkv_poseidon* ps, const uint8_t* proofs, size_t n, size_t stride, const uint8_t* lead,
                              size_t n_lead, const uint32_t* layout, size_t L, const uint32_t* point_offsets, size_t P,
                              const uint32_t* seg_len, size_t S, uint8_t* challenges, uint8_t* points64, uint8_t* ok);

/* ---- multi-GPU in ONE process (SURVEY.md 8b `*_multi_gpu`, 8e) -----------------
 * For a caller without torchrun -- the reference's `NativeLoader` is a unit struct
 * with static dispatch (loader.rs:108, native.rs:11-19).  A `snarkv_mgpu` owns one
 * context + HIP stream per entry of `devices`; a device may be listed several times
 * (each entry is a rank of its own: a 1-GPU box runs the 8-rank code path that way).
 * Sharding follows the reference's own chunking, `chunk = ceil(n / ranks)`
 * (util/msm.rs:311-336, GPUs in place of rayon threads); partial results travel as
 * point-to-point peer copies over xGMI (144 bytes per rank) to rank 0's device and
 * are folded there.  variant: 0 = point-sharded (every rank a full Pippenger of its
 * shard), 1 = bucket-sharded (the "bucket-sum allreduce" of BASELINE config 4: all
 * ranks fill the GLOBAL bucket grid, exchange it by window range, reduce the windows
 * they own).  Same group element, same bytes, for any rank count and either variant.
 * (One process per GPU keeps using RCCL: snark-verifier_amd/distributed.py.)
 * Threading: one call at a time per handle (its ranks' streams are driven by the
 * calling thread); different handles are independent.                            */
typedef struct snarkv_mgpu snarkv_mgpu;
int snarkv_mgpu_create(const int* devices, int n, snarkv_mgpu** out);
void snarkv_mgpu_destroy(snarkv_mgpu* mg);
int snarkv_mgpu_size(const snarkv_mgpu* mg);
snarkv_ctx* snarkv_mgpu_ctx(snarkv_mgpu* mg, int rank);              /* rank's context (to allocate / fill its shard) */
int snarkv_mgpu_shard(const snarkv_mgpu* mg, size_t n_total, int rank, size_t* lo, size_t* hi);
/* Directed pairs of DISTINCT devices of the handle: direct (xGMI) peer access enabled / not offered by the platform /
 * offered but refused.  Copies work in all three cases (the runtime stages what it must); the last refusal's text is
 * in snarkv_last_error() after snarkv_mgpu_create. */
int snarkv_mgpu_peer_access(const snarkv_mgpu* mg, int* enabled, int* unavailable, int* failed);
/* How the 144-byte partials travel: point-to-point peer copies to rank 0 + fold + broadcast of the result (default), or
 * ONE grouped RCCL all-gather over xGMI (`ncclCommInitAll` over the handle's devices, which must be distinct) after
 * which every rank folds.  Either way every rank's device holds the result afterwards (snarkv_mgpu_result_dev). */
#define SNARKV_MGPU_TRANSPORT_PEER_COPY 0
#define SNARKV_MGPU_TRANSPORT_RCCL 1
int snarkv_mgpu_set_transport(snarkv_mgpu* mg, int transport);
/* device pointer (on rank's device) of the 64-byte affine result of the handle's last MSM: all-reduce semantics */
const void* snarkv_mgpu_result_dev(const snarkv_mgpu* mg, int rank);
/* host buffers of the WHOLE MSM (sharded, staged and reduced inside); n >= 1 */
int snarkv_g1_msm_pippenger_mgpu(snarkv_mgpu* mg, const uint8_t* scalars32, const uint8_t* points64, size_t n,
                                 int variant, uint8_t out64[64]);
/* shards already resident: d_scalars32[g] / d_points64[g] on rank g's device, counts[g] points (0 allowed) */
int snarkv_g1_msm_pippenger_mgpu_dev(snarkv_mgpu* mg, const void* const* d_scalars32, const void* const* d_points64,
                                     const size_t* counts, int window_bits, int variant, uint8_t out64[64]);
/* A BATCH of `jobs` MSMs, every one sharded over the ranks, with ONE exchange for the whole batch (jobs x 144 bytes per
 * rank) -- the single-process form of what `bench.py --gpus N` times with one process per GPU.  Rank g's shard of job j:
 * d_scalars32[g * jobs + j] / d_points64[g * jobs + j] on rank g's device, counts[g * jobs + j] points (0 allowed as long
 * as every job has a point somewhere).  Each rank runs the batch pipeline of snarkv_g1_msm_pippenger_many_partial_dev
 * (enqueued by a host thread of its own), the partials travel by the handle's transport, out64s[j] = job j's affine
 * result (host memory, jobs x 64 bytes); every rank's device holds the results too (snarkv_mgpu_results_many_dev).
 * jobs <= 65 536 (SNARKV_ERR_LENGTH beyond). */
int snarkv_g1_msm_pippenger_many_mgpu_dev(snarkv_mgpu* mg, size_t jobs, const void* const* d_scalars32,
                                          const void* const* d_points64, const size_t* counts, int window_bits,
                                          uint8_t* out64s);
const void* snarkv_mgpu_results_many_dev(const snarkv_mgpu* mg, int rank);
/* decide_all with the accumulators sharded over the ranks; returns 1 iff all accepted, ok[i] per accumulator */
int snarkv_kzg_decide_batch_mgpu(snarkv_mgpu* mg, const uint8_t g1_64[64], const uint8_t g2_128[128],
                                 const uint8_t s_g2_128[128], const uint8_t* accs128, size_t m, uint8_t* ok);

/* Hint: the caller keeps SEVERAL large MSMs in flight on several contexts (one context + stream each).  The Pippenger then cuts
 * the sorted stream into longer runs per lane -- less total work per MSM (-3.5 % at 2^20 with 4 in flight) at the price of a
 * longer single-MSM latency (+4 %), because one MSM alone no longer fills every wave slot.  Same bytes either way.
 * `enabled`: 0 off; 1 on (the number of contexts in flight unknown: 16 assumed where it matters -- the batched small-MSM
 * launch sizes its lane groups by the share of the GPU it can count on); n >= 2: on, with n contexts in flight. */
int snarkv_ctx_set_throughput_hint(snarkv_ctx* ctx, int enabled);
/* points ONE launch of the Pippenger kernels processes for an n-point MSM: n itself, or the 2^20-point chunk of the
 * chunk pipeline large MSMs run as (csrc/capi.hip pippenger_maybe_split) -- what a per-launch roofline divides by */
int snarkv_g1_msm_launch_points(size_t n, size_t* per_launch);
/* ... for a call that passes `window_bits` (an explicit window size keeps the single launch whatever n is) */
int snarkv_g1_msm_launch_points_ex(size_t n, int window_bits, size_t* per_launch);
int snarkv_set_stage_timing(snarkv_ctx* ctx, int enabled);
int snarkv_get_stage_timing(snarkv_ctx* ctx, float ms[SNARKV_PIP_STAGES]);

#ifdef __cplusplus
}
#endif
#endif /* SNARKV_AMD_H */
