// C ABI of the host mirror -> libsnarkv_host.so (include/snarkv_host.h).  Thin: every function parses its byte
// arguments, calls the C++ mirror of the reference API (plonk.hpp / pcs.hpp / aggregation.hpp) and maps
// `Result<_, Error>` / panics to return codes.  All EC work happens behind GpuNativeLoader on the device.
#include "../../include/snarkv_host.h"

#include <cstdio>
#include <cstring>
#include <string>

#include "aggregation.hpp"
#include "pcs.hpp"
#include "plonk.hpp"
#include "serde_json.hpp"
#include "transcript.hpp"
#include "wire.hpp"

using namespace snarkv_host;

struct snarkv_host_protocol {
  PlonkProtocol pr;
};
struct snarkv_host_dk {
  KzgDecidingKey dk;
};
struct snarkv_host_snark {
  snarkv_host_protocol protocol;
  std::vector<uint8_t> instances, proof;
};

namespace {
// SNARKV_HOST_TRANSCRIPT_POSEIDON_AUTO: where a batch of n Poseidon proofs is hashed.  The device launch is one latency chain
// of ~3 ms whatever the batch; the host threads' time grows with n.  On the scalar sponge they cross at ~512 proofs
// (SNARKV_HOST_POSEIDON_DEVICE_MIN); with the AVX-512 IFMA sponge and the batch's points decompressed by the device as
// hints the host reads 512 proofs in ~1.2 ms and 1 024 in 2.3 ms (device: 3.3 / 3.6 ms), level end to end at 1 024 with
// the wider tail -- so there the host keeps batches below 1 024 (profiles/r05_host_poseidon.txt).
// `pipelined`: the caller is `snarkv_host_aggregate` (ONE job), whose host route overlaps the reading, the MSMs and the
// accumulation sponge from SNARKV_HOST_PIPELINE_MIN proofs on (host/aggregation.hpp `aggregate_pipelined`): the job is
// then bounded by the one thread that absorbs the accumulators whatever hashed the proofs, and the device launch's 3 ms
// would only delay its start -- as long as the pool has the threads to read a chunk faster than the sponge absorbs one.
static int poseidon_auto_route(size_t n, bool pipelined = false) {
  if (pipelined && HostPool::get().size() >= 32) {
    const size_t pmin = Aggregator<Gwc19, PoseidonTranscript>::pipeline_min();
    if (pmin && n >= pmin) return SNARKV_HOST_TRANSCRIPT_POSEIDON;
  }
  // (round 6: 5.2 us permutations and grouped point decoding -- 64 threads read 1 024 proofs in 2.3 ms, the device in 3.5)
  const size_t device_min = poseidon_ifma::available() ? 3 * (size_t)SNARKV_HOST_POSEIDON_DEVICE_MIN : (size_t)SNARKV_HOST_POSEIDON_DEVICE_MIN;
  return n >= device_min ? SNARKV_HOST_TRANSCRIPT_POSEIDON_DEVICE : SNARKV_HOST_TRANSCRIPT_POSEIDON;
}

thread_local std::string g_last_error;

int error_code(const Error& e) {
  g_last_error = e.msg;
  switch (e.kind) {
    case Error::Transcript: return SNARKV_HOST_ERR_TRANSCRIPT;
    case Error::InvalidInstances: return SNARKV_HOST_ERR_INVALID_INSTANCES;
    case Error::InvalidProtocol: return SNARKV_HOST_ERR_INVALID_PROTOCOL;
    case Error::AssertionFailure: return 0;
    default: return SNARKV_HOST_ERR_OTHER;
  }
}

template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const Panic& e) {
    g_last_error = std::string("panic: ") + e.what();
    return SNARKV_HOST_ERR_PANIC;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return SNARKV_HOST_ERR_DEVICE;
  }
}

int arg_error(const char* what) {
  g_last_error = what;
  return SNARKV_HOST_ERR_ARG;
}

std::vector<KzgAccumulator> accs_from_bytes(const uint8_t* accs128, uint32_t m) {
  std::vector<KzgAccumulator> accs;
  accs.reserve(m);
  for (uint32_t i = 0; i < m; ++i)
    accs.push_back(KzgAccumulator{G1Affine::from_bytes(accs128 + 128 * (size_t)i), G1Affine::from_bytes(accs128 + 128 * (size_t)i + 64)});
  return accs;
}

// read_proof of every proof with transcript TR; strict: no bytes may be left
template <class MOS, class TR>
Error read_all(const KzgSuccinctVerifyingKey& svk, const PlonkProtocol& pr,
               const std::vector<std::vector<std::vector<Fr>>>& insts, const std::vector<std::vector<uint8_t>>& pbytes,
               bool strict, std::vector<PlonkProof<MOS>>& pfs, bool* trailing) {
  const size_t n = pbytes.size();
  pfs.resize(n);
  std::vector<Error> errs(n);
  std::vector<uint8_t> left(n, 0);
  parallel_for(n, HostPool::get().size(), [&](size_t i) {
    TR t(pbytes[i]);
    auto pf = PlonkSuccinctVerifier<MOS>::read_proof(svk, pr, insts[i], t);
    if (!pf.ok()) {
      errs[i] = pf.err;
      return;
    }
    left[i] = t.remaining() != 0;
    pfs[i] = std::move(*pf.value);
  }, std::is_same<TR, PoseidonTranscript>::value ? 1 : 2);
  for (auto& e : errs)
    if (!e.ok()) return e;
  if (strict)
    for (uint8_t l : left)
      if (l) *trailing = true;
  return Error{};
}

template <class MOS>
int succinct_verify_batch(const PlonkProtocol& pr, const KzgDecidingKey& dk, int transcript, const uint8_t* instances,
                          size_t ilen, const uint8_t* proofs, size_t prlen, uint32_t n, bool strict,
                          std::vector<KzgAccumulator>& all) {
  std::vector<std::vector<std::vector<Fr>>> insts;
  std::vector<std::vector<uint8_t>> pbytes;
  wire::split_batch(instances, ilen, proofs, prlen, n, insts, pbytes);
  std::vector<PlonkProof<MOS>> pfs;
  bool trailing = false;
  Error e;
  if (transcript == SNARKV_HOST_TRANSCRIPT_POSEIDON_AUTO)
    transcript = poseidon_auto_route(n);
  if (transcript == SNARKV_HOST_TRANSCRIPT_EVM) {
    e = read_all<MOS, EvmTranscript>(dk.svk, pr, insts, pbytes, strict, pfs, &trailing);
  } else if (transcript == SNARKV_HOST_TRANSCRIPT_POSEIDON) {
    e = read_all<MOS, PoseidonTranscript>(dk.svk, pr, insts, pbytes, strict, pfs, &trailing);
  } else if (transcript == SNARKV_HOST_TRANSCRIPT_POSEIDON_DEVICE) {
    pfs.resize(n);
    e = Aggregator<MOS, PoseidonTranscriptOnDevice>::read_proofs_device_hashed(dk.svk, pr, insts, pbytes,
                                                                               HostPool::get().size(), pfs);
  } else {
    return arg_error("unknown transcript kind");
  }
  if (!e.ok()) return error_code(e);
  if (trailing) {
    g_last_error = "trailing bytes after a proof";
    return SNARKV_HOST_ERR_TRAILING;
  }
  std::vector<const PlonkProtocol*> prs(n, &pr);
  auto accs = PlonkSuccinctVerifier<MOS>::verify_batch(dk.svk, prs, insts, pfs, HostPool::get().size());
  if (!accs.ok()) return error_code(accs.err);
  for (auto& v : *accs.value) all.insert(all.end(), v.begin(), v.end());
  return 1;
}

template <class MOS, class TR>
int aggregate_run(const PlonkProtocol& pr, const KzgDecidingKey& dk, const std::vector<std::vector<std::vector<Fr>>>& insts,
                  const std::vector<std::vector<uint8_t>>& proofs, unsigned threads, double* timings_ms, uint8_t* acc_out) {
  AggregationTimings tm;
  KzgAccumulator acc;
  Error e = Aggregator<MOS, TR>::aggregate_and_decide(dk, pr, insts, proofs, threads, &tm, &acc);
  if (timings_ms) {
    timings_ms[0] = tm.read_proofs;
    timings_ms[1] = tm.fr_algebra;
    timings_ms[2] = tm.msm_device;
    timings_ms[3] = tm.accumulate;
    timings_ms[4] = tm.decide;
    timings_ms[5] = tm.total;
  }
  if (e.ok() || e.kind == Error::AssertionFailure) {
    if (acc_out) acc.to_bytes(acc_out);
  }
  return e.ok() ? 1 : error_code(e);
}

template <class MOS, class TR>
int aggregate_many_run(const PlonkProtocol& pr, const KzgDecidingKey& dk, const std::vector<std::vector<std::vector<Fr>>>& insts,
                       const std::vector<std::vector<uint8_t>>& proofs, const std::vector<uint32_t>& sizes, unsigned threads,
                       double* timings_ms, uint8_t* accs_out, uint8_t* ok_out) {
  AggregationTimings tm;
  auto r = Aggregator<MOS, TR>::aggregate_and_decide_many(dk, pr, insts, proofs, sizes, threads, &tm);
  if (!r.ok()) return error_code(r.err);
  if (timings_ms) {
    timings_ms[0] = tm.read_proofs;
    timings_ms[1] = tm.fr_algebra;
    timings_ms[2] = tm.msm_device;
    timings_ms[3] = tm.accumulate;
    timings_ms[4] = tm.decide;
    timings_ms[5] = tm.total;
  }
  int all = 1;
  for (size_t k = 0; k < sizes.size(); ++k) {
    if (accs_out) r.value->accs[k].to_bytes(accs_out + 128 * k);
    if (ok_out) ok_out[k] = r.value->ok[k];
    if (!r.value->ok[k]) all = 0;
  }
  return all;
}

template <class MOS>
int aggregate_many_mos(const PlonkProtocol& pr, const KzgDecidingKey& dk, int transcript, const uint8_t* instances, size_t ilen,
                       const uint8_t* proofs, size_t prlen, const uint32_t* job_sizes, uint32_t n_jobs, unsigned threads,
                       double* timings_ms, uint8_t* accs_out, uint8_t* ok_out) {
  std::vector<uint32_t> sizes(job_sizes, job_sizes + n_jobs);
  uint64_t n = 0;
  for (uint32_t k : sizes) n += k;
  if (n == 0 || n > 0xFFFFFFFFull) return arg_error("job sizes");
  std::vector<std::vector<std::vector<Fr>>> insts;
  std::vector<std::vector<uint8_t>> pbytes;
  wire::split_batch(instances, ilen, proofs, prlen, (uint32_t)n, insts, pbytes);
  if (threads == 0) threads = HostPool::get().size();
  if (transcript == SNARKV_HOST_TRANSCRIPT_POSEIDON_AUTO)
    transcript = poseidon_auto_route(n);
  switch (transcript) {
    case SNARKV_HOST_TRANSCRIPT_EVM:
      return aggregate_many_run<MOS, EvmTranscript>(pr, dk, insts, pbytes, sizes, threads, timings_ms, accs_out, ok_out);
    case SNARKV_HOST_TRANSCRIPT_POSEIDON:
      return aggregate_many_run<MOS, PoseidonTranscript>(pr, dk, insts, pbytes, sizes, threads, timings_ms, accs_out, ok_out);
    case SNARKV_HOST_TRANSCRIPT_POSEIDON_DEVICE:
      return aggregate_many_run<MOS, PoseidonTranscriptOnDevice>(pr, dk, insts, pbytes, sizes, threads, timings_ms, accs_out, ok_out);
    default: return arg_error("unknown transcript kind");
  }
}

template <class MOS>
int aggregate_mos(const PlonkProtocol& pr, const KzgDecidingKey& dk, int transcript, const uint8_t* instances, size_t ilen,
                  const uint8_t* proofs, size_t prlen, uint32_t n, unsigned threads, double* timings_ms, uint8_t* acc_out) {
  std::vector<std::vector<std::vector<Fr>>> insts;
  std::vector<std::vector<uint8_t>> pbytes;
  wire::split_batch(instances, ilen, proofs, prlen, n, insts, pbytes);
  if (threads == 0) threads = HostPool::get().size();
  if (transcript == SNARKV_HOST_TRANSCRIPT_POSEIDON_AUTO)
    transcript = poseidon_auto_route(n, true);
  switch (transcript) {
    case SNARKV_HOST_TRANSCRIPT_EVM: return aggregate_run<MOS, EvmTranscript>(pr, dk, insts, pbytes, threads, timings_ms, acc_out);
    case SNARKV_HOST_TRANSCRIPT_POSEIDON: return aggregate_run<MOS, PoseidonTranscript>(pr, dk, insts, pbytes, threads, timings_ms, acc_out);
    case SNARKV_HOST_TRANSCRIPT_POSEIDON_DEVICE:
      return aggregate_run<MOS, PoseidonTranscriptOnDevice>(pr, dk, insts, pbytes, threads, timings_ms, acc_out);
    default: return arg_error("unknown transcript kind");
  }
}

}  // namespace

namespace {
// `KzgAs::create_proof` over a fresh transcript of kind TR, then `KzgAsProof::read` of the bytes it wrote for r
template <class TR>
int kzg_as_create_proof_tr(const std::vector<KzgAccumulator>& accs, const KzgAsProvingKey& pk, const Fr& blind,
                           uint8_t acc_out[128], std::vector<uint8_t>& proof, uint8_t* r_out32) {
  TR t;
  auto acc = KzgAs<Gwc19>::create_proof(pk, accs, t, blind);
  if (!acc.ok()) return error_code(acc.err);
  acc.value->to_bytes(acc_out);
  proof = t.finalize();
  if (r_out32) {
    TR rd(proof);
    auto pf = KzgAsProof::read(KzgAsVerifyingKey{pk.zk()}, accs, rd);
    if (!pf.ok()) return error_code(pf.err);
    pf.value->r.to_bytes(r_out32);
  }
  return 1;
}
template <class TR>
int kzg_as_verify_tr(const std::vector<KzgAccumulator>& accs, bool zk, const uint8_t* proof, size_t proof_len,
                     uint8_t acc_out[128], uint8_t* r_out32) {
  TR t(std::vector<uint8_t>(proof, proof + proof_len));
  auto pf = KzgAs<Gwc19>::read_proof(KzgAsVerifyingKey{zk}, accs, t);
  if (!pf.ok()) return error_code(pf.err);
  if (t.remaining() != 0) {
    g_last_error = "trailing bytes after the accumulation proof";
    return SNARKV_HOST_ERR_TRAILING;
  }
  auto acc = KzgAs<Gwc19>::verify(KzgAsVerifyingKey{zk}, accs, *pf.value);
  if (!acc.ok()) return error_code(acc.err);
  acc.value->to_bytes(acc_out);
  if (r_out32) pf.value->r.to_bytes(r_out32);
  return 1;
}
}  // namespace

extern "C" {

const char* snarkv_host_last_error(void) { return g_last_error.c_str(); }

int snarkv_host_protocol_parse(const uint8_t* bytes, size_t len, int format, snarkv_host_protocol** out) {
  if (!bytes || !out) return arg_error("null argument");
  *out = nullptr;
  return guarded([&] {
    PlonkProtocol pr;
    if (format == SNARKV_HOST_PROTOCOL_PACKED) pr = wire::parse_protocol(bytes, len);
    else if (format == SNARKV_HOST_PROTOCOL_SERDE_JSON) pr = serde_json::parse_protocol(bytes, len);
    else if (format == SNARKV_HOST_PROTOCOL_BINCODE) pr = bincode::parse_protocol(bytes, len);
    else return arg_error("unknown protocol format");
    *out = new snarkv_host_protocol{std::move(pr)};
    return 1;
  });
}
void snarkv_host_protocol_free(snarkv_host_protocol* p) { delete p; }

int snarkv_host_protocol_pack(const snarkv_host_protocol* p, uint8_t* out, size_t cap, size_t* len_out) {
  if (!p) return arg_error("null argument");
  return guarded([&] {
    std::vector<uint8_t> b = wire::pack_protocol(p->pr);
    if (len_out) *len_out = b.size();
    if (!out || cap < b.size()) {
      g_last_error = "output buffer too small";
      return SNARKV_HOST_ERR_CAPACITY;
    }
    memcpy(out, b.data(), b.size());
    return 1;
  });
}

int snarkv_host_snark_parse(const uint8_t* bytes, size_t len, int format, snarkv_host_snark** out) {
  if (!bytes || !out) return arg_error("null argument");
  *out = nullptr;
  return guarded([&] {
    interchange::SnarkData d;
    if (format == SNARKV_HOST_PROTOCOL_SERDE_JSON) d = serde_json::parse_snark(bytes, len);
    else if (format == SNARKV_HOST_PROTOCOL_BINCODE) d = bincode::parse_snark(bytes, len);
    else return arg_error("a Snark comes as serde_json (1) or bincode (2)");
    *out = new snarkv_host_snark{snarkv_host_protocol{std::move(d.protocol)}, wire::pack_instances(d.instances), std::move(d.proof)};
    return 1;
  });
}
void snarkv_host_snark_free(snarkv_host_snark* s) { delete s; }
const snarkv_host_protocol* snarkv_host_snark_protocol(const snarkv_host_snark* s) { return s ? &s->protocol : nullptr; }
static int copy_out(const std::vector<uint8_t>& b, uint8_t* out, size_t cap, size_t* len_out) {
  if (len_out) *len_out = b.size();
  if (!out || cap < b.size()) {
    g_last_error = "output buffer too small";
    return SNARKV_HOST_ERR_CAPACITY;
  }
  if (!b.empty()) memcpy(out, b.data(), b.size());
  return 1;
}
int snarkv_host_snark_instances(const snarkv_host_snark* s, uint8_t* out, size_t cap, size_t* len_out) {
  if (!s) return arg_error("null argument");
  return copy_out(s->instances, out, cap, len_out);
}
int snarkv_host_snark_proof(const snarkv_host_snark* s, uint8_t* out, size_t cap, size_t* len_out) {
  if (!s) return arg_error("null argument");
  return copy_out(s->proof, out, cap, len_out);
}

int snarkv_host_dk_create(const uint8_t g1[64], const uint8_t g2[128], const uint8_t s_g2[128], snarkv_host_dk** out) {
  if (!g1 || !g2 || !s_g2 || !out) return arg_error("null argument");
  *out = nullptr;
  return guarded([&] {
    auto* h = new snarkv_host_dk{KzgDecidingKey(G1Affine::from_bytes(g1), G2Affine::from_bytes(g2), G2Affine::from_bytes(s_g2))};
    try {
      h->dk.handle();  // G2 line tables now: per-key setup, not per-call work
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
    return 1;
  });
}
void snarkv_host_dk_free(snarkv_host_dk* dk) { delete dk; }

int snarkv_host_plonk_succinct_verify_batch(const snarkv_host_protocol* protocol, const snarkv_host_dk* dk, int mos,
                                            int transcript, const uint8_t* instances, size_t instances_len,
                                            const uint8_t* proofs, size_t proofs_len, uint32_t n, int strict,
                                            uint8_t* accs_out, size_t accs_cap, uint32_t* n_accs) {
  if (!protocol || !dk || (n && (!instances || !proofs))) return arg_error("null argument");
  return guarded([&] {
    std::vector<KzgAccumulator> all;
    int rc = mos == SNARKV_HOST_MOS_GWC19
                 ? succinct_verify_batch<Gwc19>(protocol->pr, dk->dk, transcript, instances, instances_len, proofs, proofs_len, n, strict != 0, all)
             : mos == SNARKV_HOST_MOS_BDFG21
                 ? succinct_verify_batch<Bdfg21>(protocol->pr, dk->dk, transcript, instances, instances_len, proofs, proofs_len, n, strict != 0, all)
                 : arg_error("unknown multi-open scheme");
    if (rc != 1) return rc;
    if (n_accs) *n_accs = (uint32_t)all.size();
    if (accs_out) {
      if (128 * all.size() > accs_cap) {
        g_last_error = "accumulator buffer too small";
        return SNARKV_HOST_ERR_CAPACITY;
      }
      for (size_t i = 0; i < all.size(); ++i) all[i].to_bytes(accs_out + 128 * i);
    }
    return 1;
  });
}

int snarkv_host_kzg_as_accumulate(const uint8_t* accs128, uint32_t m, uint8_t acc_out[128], uint8_t* r_out32) {
  if (!accs128 || !acc_out) return arg_error("null argument");
  return guarded([&] {
    auto accs = accs_from_bytes(accs128, m);
    EvmTranscript t;
    // create_proof = absorb every accumulator, squeeze r, verify (accumulation.rs:148-197, non-zk)
    auto proof = KzgAs<Gwc19>::read_proof(KzgAsVerifyingKey{}, accs, t);
    if (!proof.ok()) return error_code(proof.err);
    auto acc = KzgAs<Gwc19>::verify(KzgAsVerifyingKey{}, accs, *proof.value);
    if (!acc.ok()) return error_code(acc.err);
    acc.value->to_bytes(acc_out);
    if (r_out32) proof.value->r.to_bytes(r_out32);
    return 1;
  });
}

int snarkv_host_kzg_as_create_proof(const uint8_t* accs128, uint32_t m, int transcript, const uint8_t* pk_g_sg128,
                                    const uint8_t* blind_scalar32, uint8_t acc_out[128], uint8_t* proof_out,
                                    size_t proof_cap, size_t* proof_len, uint8_t* r_out32) {
  if (!accs128 || !acc_out || (pk_g_sg128 && !blind_scalar32)) return arg_error("null argument");
  return guarded([&] {
    auto accs = accs_from_bytes(accs128, m);
    KzgAsProvingKey pk;
    Fr blind;
    if (pk_g_sg128) {  // zero-knowledge: the blind pair (s_g * b, g * b) goes into the proof (accumulation.rs:163-175)
      pk.g = std::make_pair(G1Affine::from_bytes(pk_g_sg128), G1Affine::from_bytes(pk_g_sg128 + 64));
      if (!Fr::from_bytes(blind_scalar32, &blind)) return arg_error("blind scalar is not a canonical Fr");
    }
    std::vector<uint8_t> proof;
    int rc = transcript == SNARKV_HOST_TRANSCRIPT_EVM ? kzg_as_create_proof_tr<EvmTranscript>(accs, pk, blind, acc_out, proof, r_out32)
             : transcript == SNARKV_HOST_TRANSCRIPT_POSEIDON ? kzg_as_create_proof_tr<PoseidonTranscript>(accs, pk, blind, acc_out, proof, r_out32)
                                                             : arg_error("unknown transcript kind");
    if (rc != 1) return rc;
    if (proof_len) *proof_len = proof.size();
    if (proof.size() > proof_cap || (proof.size() && !proof_out)) {
      g_last_error = "proof buffer too small";
      return SNARKV_HOST_ERR_CAPACITY;
    }
    if (proof.size()) memcpy(proof_out, proof.data(), proof.size());
    return 1;
  });
}

int snarkv_host_kzg_as_verify(const uint8_t* accs128, uint32_t m, int transcript, int zk, const uint8_t* proof,
                              size_t proof_len, uint8_t acc_out[128], uint8_t* r_out32) {
  if (!accs128 || !acc_out || (proof_len && !proof)) return arg_error("null argument");
  return guarded([&] {
    auto accs = accs_from_bytes(accs128, m);
    static const uint8_t none = 0;
    const uint8_t* pb = proof_len ? proof : &none;
    if (transcript == SNARKV_HOST_TRANSCRIPT_EVM) return kzg_as_verify_tr<EvmTranscript>(accs, zk != 0, pb, proof_len, acc_out, r_out32);
    if (transcript == SNARKV_HOST_TRANSCRIPT_POSEIDON) return kzg_as_verify_tr<PoseidonTranscript>(accs, zk != 0, pb, proof_len, acc_out, r_out32);
    return arg_error("unknown transcript kind");
  });
}

int snarkv_host_kzg_decide(const snarkv_host_dk* dk, const uint8_t acc128[128]) {
  if (!dk || !acc128) return arg_error("null argument");
  return guarded([&] {
    Error e = KzgAs<Gwc19>::decide(dk->dk, accs_from_bytes(acc128, 1)[0]);
    return e.ok() ? 1 : error_code(e);
  });
}

int snarkv_host_kzg_decide_all(const snarkv_host_dk* dk, const uint8_t* accs128, uint32_t m, uint8_t* ok_out) {
  if (!dk || (m && !accs128)) return arg_error("null argument");
  return guarded([&] {
    if (m == 0) return 1;  // decide_all of nothing is Ok(()) (decider.rs:84-93)
    std::vector<uint8_t> ok(m);
    snarkv_dk* h = dk->dk.handle();
    {
      DeviceScope lock;
      int rc = bn254_kzg_dk_decide_batch(h, accs128, m, ok.data());
      if (rc < 0) throw std::runtime_error(std::string("bn254_kzg_dk_decide_batch: ") + snarkv_last_error());
    }
    if (ok_out) memcpy(ok_out, ok.data(), m);
    for (uint8_t o : ok)
      if (!o) return error_code(Error::assertion("e(lhs, g2) e(rhs, -s_g2) == O"));
    return 1;
  });
}

int snarkv_host_aggregate(const snarkv_host_protocol* protocol, const snarkv_host_dk* dk, int mos, int transcript,
                          const uint8_t* instances, size_t instances_len, const uint8_t* proofs, size_t proofs_len,
                          uint32_t n, unsigned host_threads, double* timings_ms, uint8_t* acc_out) {
  if (!protocol || !dk || !instances || !proofs) return arg_error("null argument");
  return guarded([&] {
    if (mos == SNARKV_HOST_MOS_GWC19)
      return aggregate_mos<Gwc19>(protocol->pr, dk->dk, transcript, instances, instances_len, proofs, proofs_len, n, host_threads, timings_ms, acc_out);
    if (mos == SNARKV_HOST_MOS_BDFG21)
      return aggregate_mos<Bdfg21>(protocol->pr, dk->dk, transcript, instances, instances_len, proofs, proofs_len, n, host_threads, timings_ms, acc_out);
    return arg_error("unknown multi-open scheme");
  });
}

int snarkv_host_aggregate_many(const snarkv_host_protocol* protocol, const snarkv_host_dk* dk, int mos, int transcript,
                               const uint8_t* instances, size_t instances_len, const uint8_t* proofs, size_t proofs_len,
                               const uint32_t* job_sizes, uint32_t n_jobs, unsigned host_threads, double* timings_ms,
                               uint8_t* accs_out, uint8_t* ok_out) {
  if (!protocol || !dk || !instances || !proofs || !job_sizes || n_jobs == 0) return arg_error("null argument");
  return guarded([&] {
    if (mos == SNARKV_HOST_MOS_GWC19)
      return aggregate_many_mos<Gwc19>(protocol->pr, dk->dk, transcript, instances, instances_len, proofs, proofs_len, job_sizes,
                                       n_jobs, host_threads, timings_ms, accs_out, ok_out);
    if (mos == SNARKV_HOST_MOS_BDFG21)
      return aggregate_many_mos<Bdfg21>(protocol->pr, dk->dk, transcript, instances, instances_len, proofs, proofs_len, job_sizes,
                                        n_jobs, host_threads, timings_ms, accs_out, ok_out);
    return arg_error("unknown multi-open scheme");
  });
}

int snarkv_host_plonk_verify(const snarkv_host_protocol* protocol, const snarkv_host_dk* dk, int mos, int transcript,
                             const uint8_t* instances, size_t instances_len, const uint8_t* proofs, size_t proofs_len,
                             uint32_t n) {
  if (!protocol || !dk || (n && (!instances || !proofs))) return arg_error("null argument");
  return guarded([&] {
    std::vector<KzgAccumulator> all;
    int rc = mos == SNARKV_HOST_MOS_GWC19
                 ? succinct_verify_batch<Gwc19>(protocol->pr, dk->dk, transcript, instances, instances_len, proofs, proofs_len, n, false, all)
             : mos == SNARKV_HOST_MOS_BDFG21
                 ? succinct_verify_batch<Bdfg21>(protocol->pr, dk->dk, transcript, instances, instances_len, proofs, proofs_len, n, false, all)
                 : arg_error("unknown multi-open scheme");
    if (rc != 1) return rc;
    Error e = KzgAs<Gwc19>::decide_all(dk->dk, all);
    return e.ok() ? 1 : error_code(e);
  });
}

int snarkv_host_accumulator_to_limbs(const uint8_t acc128[128], uint8_t limbs_out[16 * 32]) {
  if (!acc128 || !limbs_out) return arg_error("null argument");
  return guarded([&] {
    auto limbs = LimbsEncoding<4, 68>::to_limbs(accs_from_bytes(acc128, 1)[0]);
    for (size_t i = 0; i < limbs.size(); ++i) limbs[i].to_bytes(limbs_out + 32 * i);
    return 1;
  });
}

int snarkv_host_accumulator_from_limbs(const uint8_t limbs[16 * 32], uint8_t acc_out[128]) {
  if (!limbs || !acc_out) return arg_error("null argument");
  return guarded([&] {
    Fr v[16];
    std::vector<const Fr*> ptrs;
    for (int i = 0; i < 16; ++i) {
      if (!Fr::from_bytes(limbs + 32 * i, &v[i])) throw Panic("non-canonical limb");
      ptrs.push_back(&v[i]);
    }
    auto acc = LimbsEncoding<4, 68>::from_repr(ptrs);
    if (!acc.ok()) return error_code(acc.err);
    acc.value->to_bytes(acc_out);
    return 1;
  });
}

}  // extern "C"
