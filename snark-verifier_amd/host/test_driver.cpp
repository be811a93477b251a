// Test driver for the C++ host mirror -> libsnarkv_hosttest.so (TEST HOOKS ONLY: the product API of the mirror is
// host/capi.cpp -> libsnarkv_host.so, include/snarkv_host.h): exposes scenario functions with plain
// byte buffers so the pytest suite can drive `Msm`, `KzgAs<Gwc19|Bdfg21>`,
// `LimbsEncoding` and the decider exactly as the reference's callers do
// (snark-verifier/examples/evm-verifier-with-accumulator.rs:357-380) and compare
// with oracle/kzg.py.  Links against libsnarkv_amd.so (the HIP path).
#include <chrono>
#include <cstdio>
#include <cstring>

#include "aggregation.hpp"
#include "pcs.hpp"
#include "plonk.hpp"
#include "wire.hpp"
#include "transcript.hpp"

using namespace snarkv_host;

namespace {

#include "driver_parse.inc"

// Deterministic stand-in for the hash transcripts (out of the hot path): the
// test supplies the challenges / points it wants "read".
struct ScriptedTranscript : Transcript {
  std::vector<Fr> challenges;
  std::vector<G1Affine> points;
  size_t ci = 0, pi = 0;
  std::vector<G1Affine> absorbed, written;
  Fr squeeze_challenge() override { return challenges.at(ci++); }
  Error common_ec_point(const G1Affine& p) override {
    absorbed.push_back(p);
    return {};
  }
  Error common_scalar(const Fr&) override { return {}; }
  Result<G1Affine> read_ec_point() override {
    if (pi >= points.size()) return Result<G1Affine>::Err(Error{Error::Transcript, "eof"});
    return Result<G1Affine>::Ok(points[pi++]);
  }
  Result<Fr> read_scalar() override { return Result<Fr>::Err(Error{Error::Transcript, "eof"}); }
  Error write_ec_point(const G1Affine& p) override {
    written.push_back(p);
    return {};
  }
  Error write_scalar(const Fr&) override { return {}; }
};

int error_code(const Error& e) {
  switch (e.kind) {
    case Error::Transcript: return -10;
    case Error::InvalidInstances: return -11;
    case Error::InvalidProtocol: return -12;
    case Error::AssertionFailure: return 0;
    default: return -13;
  }
}

int guarded(const std::function<int()>& f) {
  try {
    return f();
  } catch (const Panic& e) {
    fprintf(stderr, "panic: %s\n", e.what());
    return -100;
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return -101;
  }
}

}  // namespace

#include <functional>

extern "C" {

// Fr self-test hooks
void hd_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fr x, y;
  Fr::from_bytes(a, &x);
  Fr::from_bytes(b, &y);
  (x * y).to_bytes(out);
}
int hd_fr_inv(const uint8_t* a, uint8_t* out) {
  Fr x, y;
  Fr::from_bytes(a, &x);
  if (!x.invert(&y)) return 0;
  y.to_bytes(out);
  return 1;
}

// the exponentiation form of the same inverse (the cross-check of the Euclid-based `invert`)
int hd_fr_inv_fermat(const uint8_t* a, uint8_t* out) {
  Fr x, y;
  Fr::from_bytes(a, &x);
  if (!x.invert_fermat(&y)) return 0;
  y.to_bytes(out);
  return 1;
}

// Msm::evaluate through the loader (msm.rs:81-98): in = g(64) commitments-format with ONE Msm
int hd_msm_evaluate(const uint8_t* in, int with_gen, uint8_t* out64) {
  return guarded([&] {
    Reader rd{in};
    G1Affine g = rd.g1();
    std::vector<std::vector<G1Affine>> store;
    std::vector<MsmT> ms;
    read_commitments(rd, store, ms);
    G1Affine r = ms.at(0).evaluate(with_gen ? std::optional<G1Affine>(g) : std::nullopt);
    memcpy(out64, r.b, 64);
    return 0;
  });
}

// Gwc19: in = g | commitments | z | queries | v | nws | ws | u ; out = lhs||rhs ; sizes[2] = MSM term counts
int hd_gwc19_verify(const uint8_t* in, uint8_t* out128, uint32_t* sizes) {
  return guarded([&] {
    Reader rd{in};
    KzgSuccinctVerifyingKey svk{rd.g1()};
    std::vector<std::vector<G1Affine>> store;
    std::vector<MsmT> commitments;
    read_commitments(rd, store, commitments);
    Fr z = rd.fr();
    auto queries = read_queries(rd);
    Gwc19Proof proof;
    proof.v = rd.fr();
    uint32_t nw = rd.u32();
    for (uint32_t i = 0; i < nw; ++i) proof.ws.push_back(rd.g1());
    proof.u = rd.fr();
    auto ms = gwc19::msms(commitments, z, queries, proof);
    sizes[0] = (uint32_t)ms.first.pairs(svk.g).size();
    sizes[1] = (uint32_t)ms.second.pairs(svk.g).size();
    auto acc = KzgAs<Gwc19>::pcs_verify(svk, commitments, z, queries, proof);
    acc.value->to_bytes(out128);
    return 0;
  });
}

// Bdfg21: in = g | commitments | z | queries | mu | gamma | w | z_prime | w_prime
int hd_bdfg21_verify(const uint8_t* in, uint8_t* out128, uint32_t* sizes) {
  return guarded([&] {
    Reader rd{in};
    KzgSuccinctVerifyingKey svk{rd.g1()};
    std::vector<std::vector<G1Affine>> store;
    std::vector<MsmT> commitments;
    read_commitments(rd, store, commitments);
    Fr z = rd.fr();
    auto queries = read_queries(rd);
    Bdfg21Proof proof;
    proof.mu = rd.fr();
    proof.gamma = rd.fr();
    proof.w = rd.g1();
    proof.z_prime = rd.fr();
    proof.w_prime = rd.g1();
    auto ms = bdfg21::msms(commitments, z, queries, proof);
    sizes[0] = (uint32_t)ms.first.pairs(svk.g).size();
    sizes[1] = (uint32_t)ms.second.pairs(svk.g).size();
    auto acc = KzgAs<Bdfg21>::pcs_verify(svk, commitments, z, queries, proof);
    acc.value->to_bytes(out128);
    return 0;
  });
}

// KzgAs::read_proof + verify (accumulation.rs:30-63): m accumulators, challenge r,
// optional blind pair read from the (scripted) transcript.
int hd_kzg_as_verify(const uint8_t* accs128, uint32_t m, const uint8_t* r32, const uint8_t* blind128_or_null,
                     uint8_t* out128) {
  return guarded([&] {
    std::vector<KzgAccumulator> instances;
    for (uint32_t i = 0; i < m; ++i)
      instances.push_back(KzgAccumulator{G1Affine::from_bytes(accs128 + 128 * i), G1Affine::from_bytes(accs128 + 128 * i + 64)});
    ScriptedTranscript t;
    Fr r;
    if (!Fr::from_bytes(r32, &r)) return -3;
    t.challenges.push_back(r);
    KzgAsVerifyingKey vk{blind128_or_null != nullptr};
    if (blind128_or_null) {
      t.points.push_back(G1Affine::from_bytes(blind128_or_null));
      t.points.push_back(G1Affine::from_bytes(blind128_or_null + 64));
    }
    auto proof = KzgAs<Gwc19>::read_proof(vk, instances, t);
    if (!proof.ok()) return -4;
    if (t.absorbed.size() != 2 * m) return -5;  // every lhs/rhs absorbed (accumulation.rs:124-127)
    auto acc = KzgAs<Gwc19>::verify(vk, instances, *proof.value);
    acc.value->to_bytes(out128);
    return 0;
  });
}

// prover twin (accumulation.rs:148-197), zk: pk = (g, s*g), blind scalar given
int hd_kzg_as_create_proof(const uint8_t* accs128, uint32_t m, const uint8_t* r32, const uint8_t* pk128_or_null,
                           const uint8_t* blind_scalar32, uint8_t* out128, uint8_t* written128) {
  return guarded([&] {
    std::vector<KzgAccumulator> instances;
    for (uint32_t i = 0; i < m; ++i)
      instances.push_back(KzgAccumulator{G1Affine::from_bytes(accs128 + 128 * i), G1Affine::from_bytes(accs128 + 128 * i + 64)});
    ScriptedTranscript t;
    Fr r, bs;
    Fr::from_bytes(r32, &r);
    t.challenges.push_back(r);
    KzgAsProvingKey pk;
    if (pk128_or_null) {
      pk.g = std::make_pair(G1Affine::from_bytes(pk128_or_null), G1Affine::from_bytes(pk128_or_null + 64));
      Fr::from_bytes(blind_scalar32, &bs);
    }
    auto acc = KzgAs<Bdfg21>::create_proof(pk, instances, t, bs);
    acc.value->to_bytes(out128);
    if (pk128_or_null && written128) {
      memcpy(written128, t.written.at(0).b, 64);
      memcpy(written128 + 64, t.written.at(1).b, 64);
    }
    return 0;
  });
}

// decide / decide_all (decider.rs:70-93): returns 1 Ok(()), 0 Err(AssertionFailure)
int hd_decide_all(const uint8_t* g1, const uint8_t* g2, const uint8_t* s_g2, const uint8_t* accs128, uint32_t m,
                  int one_by_one) {
  return guarded([&] {
    KzgDecidingKey dk(G1Affine::from_bytes(g1), G2Affine::from_bytes(g2), G2Affine::from_bytes(s_g2));
    std::vector<KzgAccumulator> accs;
    for (uint32_t i = 0; i < m; ++i)
      accs.push_back(KzgAccumulator{G1Affine::from_bytes(accs128 + 128 * i), G1Affine::from_bytes(accs128 + 128 * i + 64)});
    if (one_by_one) {
      for (auto& a : accs)
        if (!KzgAs<Gwc19>::decide(dk, a).ok()) return 0;
      return 1;
    }
    return KzgAs<Gwc19>::decide_all(dk, accs).ok() ? 1 : 0;
  });
}

// LimbsEncoding<4,68>: 16 Fr limbs -> accumulator -> 16 limbs again
int hd_limbs_roundtrip(const uint8_t* limbs16x32, uint8_t* out128, uint8_t* limbs_out16x32) {
  return guarded([&] {
    std::vector<Fr> ls(16);
    std::vector<const Fr*> refs;
    for (int i = 0; i < 16; ++i) {
      if (!Fr::from_bytes(limbs16x32 + 32 * i, &ls[i])) return -3;
      refs.push_back(&ls[i]);
    }
    auto acc = LimbsEncoding<4, 68>::from_repr(refs);
    acc.value->to_bytes(out128);
    auto back = LimbsEncoding<4, 68>::to_limbs(*acc.value);
    for (int i = 0; i < 16; ++i) back[i].to_bytes(limbs_out16x32 + 32 * i);
    return 0;
  });
}

// ---- Keccak / EVM transcript (transcript.hpp; reference system/halo2/transcript/evm.rs)
void hd_keccak256(const uint8_t* data, size_t len, uint8_t* out32) { keccak::keccak256(data, len, out32); }

// Runs a scripted sequence of transcript operations on `EvmTranscript(proof)`.
// script: op bytes, each followed by its LE payload:
//   1 squeeze -> out += challenge(32 LE)      2 common_scalar [32]       3 common_ec_point [64]
//   4 read_scalar -> out += 32 LE             5 read_ec_point -> out += 64 LE
//   6 write_scalar [32]                       7 write_ec_point [64]
//   8 finalize -> out += len(u32) || stream
// Returns 0, or 1000 + index of the operation that returned Error::Transcript.
int hd_transcript_script(int kind, const uint8_t* script, size_t script_len, const uint8_t* proof, size_t proof_len,
                         uint8_t* out, size_t out_cap, size_t* out_len) {
  return guarded([&] {
    EvmTranscript te(kind == 0 ? std::vector<uint8_t>(proof, proof + proof_len) : std::vector<uint8_t>());
    PoseidonTranscript tp(kind != 0 ? std::vector<uint8_t>(proof, proof + proof_len) : std::vector<uint8_t>());
    if (kind == 2) tp.sponge().set_eager(true);  // kind 2: the Poseidon transcript with the eager sponge (same bytes as kind 1)
    Transcript& t = kind == 0 ? static_cast<Transcript&>(te) : static_cast<Transcript&>(tp);
    std::vector<uint8_t> o;
    size_t i = 0;
    int opi = 0;
    int rc = 0;
    while (i < script_len && rc == 0) {
      uint8_t op = script[i++];
      Error e;
      switch (op) {
        case 1: {
          uint8_t b[32];
          t.squeeze_challenge().to_bytes(b);
          o.insert(o.end(), b, b + 32);
          break;
        }
        case 2: {
          Fr x;
          if (!Fr::from_bytes(script + i, &x)) return -3;
          i += 32;
          e = t.common_scalar(x);
          break;
        }
        case 3: {
          e = t.common_ec_point(G1Affine::from_bytes(script + i));
          i += 64;
          break;
        }
        case 4: {
          auto r = t.read_scalar();
          if (!r.ok()) {
            e = r.err;
          } else {
            uint8_t b[32];
            r.value->to_bytes(b);
            o.insert(o.end(), b, b + 32);
          }
          break;
        }
        case 5: {
          auto r = t.read_ec_point();
          if (!r.ok()) {
            e = r.err;
          } else {
            o.insert(o.end(), r.value->b, r.value->b + 64);
          }
          break;
        }
        case 6: {
          Fr x;
          if (!Fr::from_bytes(script + i, &x)) return -3;
          i += 32;
          e = t.write_scalar(x);
          break;
        }
        case 7: {
          e = t.write_ec_point(G1Affine::from_bytes(script + i));
          i += 64;
          break;
        }
        case 8: {
          auto st = kind == 0 ? te.stream() : tp.stream();
          uint32_t n = (uint32_t)st.size();
          uint8_t nb[4];
          memcpy(nb, &n, 4);
          o.insert(o.end(), nb, nb + 4);
          o.insert(o.end(), st.begin(), st.end());
          break;
        }
        default:
          return -2;
      }
      if (!e.ok()) rc = 1000 + opi;
      ++opi;
    }
    if (o.size() > out_cap) return -6;
    memcpy(out, o.data(), o.size());
    *out_len = o.size();
    return rc;
  });
}

// The provenance record of a Poseidon transcript (transcript.hpp `record_layout`: what `snarkv_poseidon_read_batch` is told
// about a batch of proofs).  Runs ops 1-5 of the script above on a recording transcript, then rebuilds every absorbed
// element from its layout code -- lead value / the proof's scalar at that byte / coordinate of that point mod r -- and
// compares with what the sponge was actually given.  out: n_elems (u32) || layout codes (u32 each) || n_points (u32) ||
// point offsets (u32 each) || n_segments (u32) || segment lengths.  Returns 0, 1000 + op index for a transcript error,
// 2000 + k if element k does not rebuild.
int hd_poseidon_layout_script(const uint8_t* script, size_t script_len, const uint8_t* proof, size_t proof_len, uint8_t* out,
                              size_t out_cap, size_t* out_len) {
  return guarded([&] {
    using TR = PoseidonTranscriptT<RecordingSponge>;
    TR t(std::vector<uint8_t>(proof, proof + proof_len));
    t.record_layout();
    size_t i = 0;
    int opi = 0;
    while (i < script_len) {
      uint8_t op = script[i++];
      Error e;
      switch (op) {
        case 1: t.squeeze_challenge(); break;
        case 2: {
          Fr x;
          if (!Fr::from_bytes(script + i, &x)) return -3;
          i += 32;
          e = t.common_scalar(x);
          break;
        }
        case 3: e = t.common_ec_point(G1Affine::from_bytes(script + i)), i += 64; break;
        case 4: {
          auto r = t.read_scalar();
          if (!r.ok()) e = r.err;
          break;
        }
        case 5: {
          auto r = t.read_ec_point();
          if (!r.ok()) e = r.err;
          break;
        }
        default: return -2;
      }
      if (!e.ok()) return 1000 + opi;
      ++opi;
    }
    const auto& lay = t.layout();
    const auto& el = t.sponge().elems;
    if (lay.size() != el.size()) return -7;
    for (size_t k = 0; k < lay.size(); ++k) {
      const uint32_t kind = lay[k] >> 28, v = lay[k] & 0x0FFFFFFFu;
      Fr x;
      if (kind == TR::kSrcLead) {
        if (v >= t.lead_values().size()) return 2000 + (int)k;
        x = t.lead_values()[v];
      } else if (kind == TR::kSrcScalar) {
        if ((size_t)v + 32 > proof_len || !Fr::from_bytes(proof + v, &x)) return 2000 + (int)k;
      } else {
        if (v >= t.decoded_points().size()) return 2000 + (int)k;
        uint64_t w[4];
        memcpy(w, t.decoded_points()[v].b + (kind == TR::kSrcPy ? 32 : 0), 32);
        x = grain::fr_from_words_mod_r(w);
      }
      if (!(x == el[k])) return 2000 + (int)k;
    }
    std::vector<uint32_t> o;
    o.push_back((uint32_t)lay.size());
    o.insert(o.end(), lay.begin(), lay.end());
    o.push_back((uint32_t)t.point_offsets().size());
    for (size_t q : t.point_offsets()) o.push_back((uint32_t)q);
    std::vector<uint32_t> seg = t.sponge().seg_len;
    o.push_back((uint32_t)seg.size());
    o.insert(o.end(), seg.begin(), seg.end());
    if (o.size() * 4 > out_cap) return -6;
    memcpy(out, o.data(), o.size() * 4);
    *out_len = o.size() * 4;
    return 0;
  });
}

// The host pool (loader.hpp HostPool): every item exactly once for many (n, threads, grain) shapes, nested fan-out runs
// inline, the first exception reaches the caller and the pool survives it, several submitting threads take turns.
// Returns 0, or the number of the check that failed.
int hd_pool_selftest(int rounds) {
  return guarded([&] {
    for (int rep = 0; rep < rounds; ++rep) {
      const size_t n = 1 + (size_t)(rep * 7919) % 3000;
      const unsigned th = 1 + rep % 70;
      std::vector<std::atomic<int>> v(n);
      for (auto& x : v) x = 0;
      parallel_for(n, th, [&](size_t i) {
        v[i]++;
        if ((i & 255) == 0) parallel_for(4, 8, [&](size_t) {}, 1);
      }, 1 + rep % 5);
      for (size_t i = 0; i < n; ++i)
        if (v[i] != 1) return 1;
    }
    int caught = 0;
    for (int rep = 0; rep < 50; ++rep) {
      try {
        parallel_for(500, 64, [&](size_t i) {
          if (i == (size_t)(rep * 7) % 500) throw std::runtime_error("task failed");
        }, 1);
      } catch (const std::runtime_error&) {
        ++caught;
      }
    }
    if (caught != 50) return 2;
    std::vector<std::thread> subs;
    std::atomic<long> total{0};
    for (int t = 0; t < 4; ++t)
      subs.emplace_back([&] {
        for (int r = 0; r < 100; ++r) parallel_for(200, 16, [&](size_t) { total++; }, 1);
      });
    for (auto& t : subs) t.join();
    if (total != 4 * 100 * 200) return 3;
    bool refused = false;  // a pool task must not open a device scope (tasks of the pool stay host-only)
    parallel_for(64, 8, [&](size_t i) {
      if (i == 5) {
        try {
          DeviceScope probe;
        } catch (const std::logic_error&) {
          refused = true;
        }
      }
    }, 1);
    return refused ? 0 : 4;
  });
}

// The hint policy of the Poseidon transcript's `read_ec_point`: one compressed point in `proof32`, a device
// answer (`hint64`, `hint_ok`) beside it.  Returns 1 the point was read (from the hint or by the host's own decoding),
// 0 Error::Transcript.  `strict` = the caller's challenges were hashed over the hint (the fused device route): a finite
// point the host decodes where the hint was unusable must then be REFUSED, never silently accepted.
int hd_poseidon_hint_policy(const uint8_t* proof32, const uint8_t* hint64, int hint_ok, int strict, uint8_t* out64) {
  return guarded([&] {
    PoseidonTranscriptT<ReplaySponge> t(std::vector<uint8_t>(proof32, proof32 + 32));
    const uint8_t okb = (uint8_t)(hint_ok ? 1 : 0);
    t.set_point_hints(hint64, &okb, 1, strict != 0);
    auto p = t.read_ec_point();
    if (!p.ok()) return 0;
    memcpy(out64, p.value->b, 64);
    return 1;
  });
}

int hd_evm_transcript_script(const uint8_t* script, size_t script_len, const uint8_t* proof, size_t proof_len,
                             uint8_t* out, size_t out_cap, size_t* out_len) {
  return hd_transcript_script(0, script, script_len, proof, proof_len, out, out_cap, out_len);
}

// KzgAs over the REAL transcript, prover then verifier (accumulation.rs:148-197
// then :114-137 + :41-63): the prover absorbs the instances, writes the blind
// pair (zk) and squeezes r; the verifier re-derives r from the proof bytes.
// out: proof_len(u32) || proof || prover_acc(128) || verifier_acc(128) || r(32)
}  // extern "C" (a template cannot have C linkage)
template <class TR>
static int kzg_as_roundtrip(const uint8_t* accs128, uint32_t m, const uint8_t* pk128_or_null,
                            const uint8_t* blind_scalar32, uint8_t* out, size_t out_cap, size_t* out_len) {
  return guarded([&] {
    std::vector<KzgAccumulator> instances;
    for (uint32_t i = 0; i < m; ++i)
      instances.push_back(KzgAccumulator{G1Affine::from_bytes(accs128 + 128 * i), G1Affine::from_bytes(accs128 + 128 * i + 64)});
    KzgAsProvingKey pk;
    Fr bs;
    if (pk128_or_null) {
      pk.g = std::make_pair(G1Affine::from_bytes(pk128_or_null), G1Affine::from_bytes(pk128_or_null + 64));
      if (!Fr::from_bytes(blind_scalar32, &bs)) return -3;
    }
    TR wt;
    auto acc_p = KzgAs<Gwc19>::create_proof(pk, instances, wt, bs);
    if (!acc_p.ok()) return -4;
    std::vector<uint8_t> proof = wt.finalize();
    TR rt(proof);
    KzgAsVerifyingKey vk{pk128_or_null != nullptr};
    auto pr = KzgAs<Gwc19>::read_proof(vk, instances, rt);
    if (!pr.ok()) return -5;
    if (rt.remaining() != 0) return -7;
    auto acc_v = KzgAs<Gwc19>::verify(vk, instances, *pr.value);
    std::vector<uint8_t> o;
    uint32_t n = (uint32_t)proof.size();
    uint8_t nb[4];
    memcpy(nb, &n, 4);
    o.insert(o.end(), nb, nb + 4);
    o.insert(o.end(), proof.begin(), proof.end());
    uint8_t a[128];
    acc_p.value->to_bytes(a);
    o.insert(o.end(), a, a + 128);
    acc_v.value->to_bytes(a);
    o.insert(o.end(), a, a + 128);
    uint8_t rb[32];
    pr.value->r.to_bytes(rb);
    o.insert(o.end(), rb, rb + 32);
    if (o.size() > out_cap) return -6;
    memcpy(out, o.data(), o.size());
    *out_len = o.size();
    return 0;
  });
}

extern "C" {
int hd_kzg_as_evm_roundtrip(const uint8_t* accs128, uint32_t m, const uint8_t* pk128_or_null,
                            const uint8_t* blind_scalar32, uint8_t* out, size_t out_cap, size_t* out_len) {
  return kzg_as_roundtrip<EvmTranscript>(accs128, m, pk128_or_null, blind_scalar32, out, out_cap, out_len);
}
int hd_kzg_as_poseidon_roundtrip(const uint8_t* accs128, uint32_t m, const uint8_t* pk128_or_null,
                                 const uint8_t* blind_scalar32, uint8_t* out, size_t out_cap, size_t* out_len) {
  return kzg_as_roundtrip<PoseidonTranscript>(accs128, m, pk128_or_null, blind_scalar32, out, out_cap, out_len);
}
// round constants and MDS of the Poseidon instance (t, r_f, r_p): rc || mds, 32-byte LE each
int hd_poseidon_spec(int t, int r_f, int r_p, uint8_t* out, size_t out_cap, size_t* out_len) {
  return guarded([&] {
    const PoseidonSpec& sp = poseidon_spec(t, r_f, r_p);
    size_t n = sp.rc.size() + sp.mds.size();
    if (32 * n > out_cap) return -6;
    size_t k = 0;
    for (auto& x : sp.rc) x.to_bytes(out + 32 * k++);
    for (auto& x : sp.mds) x.to_bytes(out + 32 * k++);
    *out_len = 32 * n;
    return 0;
  });
}
// permutation of `t` words (in/out 32-byte LE each): plain = 1 the textbook rounds, 0 the optimised schedule
int hd_poseidon_permute2(int t, int r_f, int r_p, int plain, uint8_t* state) {
  return guarded([&] {
    std::vector<Fr> st((size_t)t);
    for (int i = 0; i < t; ++i)
      if (!Fr::from_bytes(state + 32 * i, &st[i])) return -3;
    if (plain) poseidon_permute_plain(st, poseidon_spec(t, r_f, r_p));
    else poseidon_permute(st, poseidon_spec(t, r_f, r_p));
    for (int i = 0; i < t; ++i) st[i].to_bytes(state + 32 * i);
    return 0;
  });
}
// up to 8 compressed points decoded together on AVX-512 IFMA (transcript.hpp g1_decompress_x8): 0 ok, 1 = no IFMA here
int hd_g1_decompress_x8(const uint8_t* enc32s, size_t n, uint8_t* out64s, uint8_t* ok_out) {
  return guarded([&] {
    const uint8_t* enc[8];
    for (size_t i = 0; i < n && i < 8; ++i) enc[i] = enc32s + 32 * i;
    G1Affine pts[8];
    uint8_t ok[8];
    if (!g1_decompress_x8(enc, n, pts, ok)) return 1;
    for (size_t i = 0; i < n; ++i) {
      ok_out[i] = ok[i];
      memcpy(out64s + 64 * i, ok[i] ? pts[i].b : G1Affine().b, 64);
    }
    return 0;
  });
}

// nanoseconds per DEPENDENT lane-wise Montgomery product on this CPU (poseidon_ifma.hpp mul_chain), or -1 without IFMA
double hd_ifma_mul_latency_ns(size_t n) {
#if defined(__x86_64__) && defined(__GNUC__)
  const poseidon_ifma::Tables* T = poseidon_ifma_tables(5, 8, 60);
  if (!T) return -1.0;
  poseidon_ifma::V x = T->one;
  auto t0 = std::chrono::steady_clock::now();
  poseidon_ifma::mul_chain(x, T->p, T->np, n);
  double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
  volatile uint64_t sink = x.l[0][0];
  (void)sink;
  return ns / (double)n;
#else
  (void)n;
  return -1.0;
#endif
}

// selects the partial-round form of the IFMA permutation (3 = default, 4 = the form it replaced); returns the previous one
int hd_poseidon_ifma_form(int form) {
#if defined(__x86_64__) && defined(__GNUC__)
  int prev = poseidon_ifma::partial_round_form();
  if (form == 3 || form == 4) poseidon_ifma::partial_round_form() = form;
  return prev;
#else
  (void)form;
  return 0;
#endif
}

// the same permutation on the AVX-512 IFMA path (host/poseidon_ifma.hpp): 0 ok, 1 = this CPU has no IFMA (state untouched)
int hd_poseidon_permute_ifma(int t, int r_f, int r_p, uint8_t* state) {
  return guarded([&] {
    std::vector<Fr> st((size_t)t);
    for (int i = 0; i < t; ++i)
      if (!Fr::from_bytes(state + 32 * i, &st[i])) return -3;
    if (!poseidon_permute_ifma(st, r_f, r_p)) return 1;
    for (int i = 0; i < t; ++i) st[i].to_bytes(state + 32 * i);
    return 0;
  });
}
int hd_poseidon_permute(int t, int r_f, int r_p, uint8_t* state) {
  return guarded([&] {
    std::vector<Fr> st((size_t)t);
    for (int i = 0; i < t; ++i)
      if (!Fr::from_bytes(state + 32 * i, &st[i])) return -3;
    poseidon_permute(st, poseidon_spec(t, r_f, r_p));
    for (int i = 0; i < t; ++i) st[i].to_bytes(state + 32 * i);
    return 0;
  });
}
}
// `PlonkVerifier::{read_proof, verify}` (verifier/plonk.rs:94-147) on N proofs of ONE protocol.
//   mos: 0 Gwc19, 1 Bdfg21;  tkind: 0 EvmTranscript, 1 PoseidonTranscript
//   proofs: N x (u32 len || bytes);  instances: N x packed instances
//   dk: g1(64) || g2(128) || s_g2(128)
// Succinct part: ONE segmented MSM launch for all proofs (`verify_batch`); then ONE `decide_all`.
// accs_out (if non-null): every accumulator (new one first, then the old ones of each proof), 128 B each.
// Returns 1 accept, 0 reject (decide failed), -10 Transcript, -11 InvalidInstances, -12 InvalidProtocol.
template <class MOS>
static int plonk_verify_impl(int tkind, const uint8_t* protocol, size_t plen, const uint8_t* instances, size_t ilen,
                             const uint8_t* proofs, size_t prlen, uint32_t n, const uint8_t* dk320, uint8_t* accs_out,
                             size_t accs_cap, uint32_t* n_accs, bool decide = true) {
  return guarded([&] {
    PlonkProtocol pr = parse_protocol(protocol, plen);
    KzgDecidingKey dk(G1Affine::from_bytes(dk320), G2Affine::from_bytes(dk320 + 64), G2Affine::from_bytes(dk320 + 192));
    std::vector<std::vector<std::vector<Fr>>> insts;
    std::vector<PlonkProof<MOS>> pfs;
    std::vector<const PlonkProtocol*> prs;
    const uint8_t* ip = instances;
    const uint8_t* pp = proofs;
    for (uint32_t i = 0; i < n; ++i) {
      // instances: each packed block is self-delimiting (count, then per column count + values)
      PReader rd{ip, instances + ilen};
      uint32_t cols = rd.u32();
      for (uint32_t c = 0; c < cols; ++c) {
        uint32_t m = rd.u32();
        rd.need(32 * (size_t)m);
        rd.p += 32 * (size_t)m;
      }
      insts.push_back(parse_instances(ip, (size_t)(rd.p - ip)));
      ip = rd.p;
      if ((size_t)(proofs + prlen - pp) < 4) throw Panic("truncated proofs");
      uint32_t len;
      memcpy(&len, pp, 4);
      pp += 4;
      if ((size_t)(proofs + prlen - pp) < (size_t)len) throw Panic("proof length runs past the buffer");
      std::vector<uint8_t> bytes(pp, pp + len);
      pp += len;
      Result<PlonkProof<MOS>> pf = Result<PlonkProof<MOS>>::Err(Error{});
      size_t remaining = 0;
      if (tkind == 0) {
        EvmTranscript t(bytes);
        pf = PlonkVerifier<MOS>::read_proof(dk, pr, insts.back(), t);
        remaining = t.remaining();
      } else {
        PoseidonTranscript t(bytes);
        pf = PlonkVerifier<MOS>::read_proof(dk, pr, insts.back(), t);
        remaining = t.remaining();
      }
      if (!pf.ok()) return error_code(pf.err);
      if (remaining != 0) return -14;  // trailing proof bytes (a test-driver check, not a reference rule)
      pfs.push_back(std::move(*pf.value));
      prs.push_back(&pr);
    }
    auto accs = PlonkSuccinctVerifier<MOS>::verify_batch(dk.svk, prs, insts, pfs);
    if (!accs.ok()) return error_code(accs.err);
    std::vector<KzgAccumulator> all;
    for (auto& v : *accs.value) all.insert(all.end(), v.begin(), v.end());
    if (n_accs) *n_accs = (uint32_t)all.size();
    if (accs_out) {
      if (128 * all.size() > accs_cap) return -6;
      for (size_t i = 0; i < all.size(); ++i) all[i].to_bytes(accs_out + 128 * i);
    }
    if (!decide) return 1;
    return KzgAs<MOS>::decide_all(dk, all).ok() ? 1 : 0;
  });
}

// `Aggregator::aggregate_and_decide` (host/aggregation.hpp) on N proofs of one protocol, inputs in the
// tests' wire format.  timings_ms[0..5] = read_proof, host algebra, device MSMs, KzgAs, decide, total.
template <class MOS, class TR>
static int aggregate_run(const PlonkProtocol& pr, const KzgDecidingKey& dk,
                         const std::vector<std::vector<std::vector<Fr>>>& insts,
                         const std::vector<std::vector<uint8_t>>& proofs, unsigned threads, double* timings_ms,
                         uint8_t* acc_out128) {
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
  if (e.ok()) {
    if (acc_out128) acc.to_bytes(acc_out128);
    return 1;
  }
  if (e.kind == Error::AssertionFailure && acc_out128) acc.to_bytes(acc_out128);
  return error_code(e);
}

template <class MOS>
static int aggregate_impl(int tkind, const uint8_t* protocol, size_t plen, const uint8_t* instances, size_t ilen,
                          const uint8_t* proofs, size_t prlen, uint32_t n, const uint8_t* dk320, unsigned threads,
                          double* timings_ms, uint8_t* acc_out128) {
  return guarded([&] {
    PlonkProtocol pr = parse_protocol(protocol, plen);
    KzgDecidingKey dk(G1Affine::from_bytes(dk320), G2Affine::from_bytes(dk320 + 64), G2Affine::from_bytes(dk320 + 192));
    std::vector<std::vector<std::vector<Fr>>> insts;
    std::vector<std::vector<uint8_t>> pbytes;
    const uint8_t* ip = instances;
    const uint8_t* pp = proofs;
    for (uint32_t i = 0; i < n; ++i) {
      PReader rd{ip, instances + ilen};
      uint32_t cols = rd.u32();
      for (uint32_t c = 0; c < cols; ++c) {
        uint32_t m = rd.u32();
        rd.need(32 * (size_t)m);
        rd.p += 32 * (size_t)m;
      }
      insts.push_back(parse_instances(ip, (size_t)(rd.p - ip)));
      ip = rd.p;
      if ((size_t)(proofs + prlen - pp) < 4) throw Panic("truncated proofs");
      uint32_t len;
      memcpy(&len, pp, 4);
      if ((size_t)(proofs + prlen - pp) < 4 + (size_t)len) throw Panic("proof length runs past the buffer");
      pbytes.emplace_back(pp + 4, pp + 4 + len);
      pp += 4 + len;
    }
    // tkind: 0 Keccak, 1 Poseidon hashed on the host, 2 Poseidon hashed on the device
    if (tkind == 0) return aggregate_run<MOS, EvmTranscript>(pr, dk, insts, pbytes, threads, timings_ms, acc_out128);
    if (tkind == 1) return aggregate_run<MOS, PoseidonTranscript>(pr, dk, insts, pbytes, threads, timings_ms, acc_out128);
    return aggregate_run<MOS, PoseidonTranscriptOnDevice>(pr, dk, insts, pbytes, threads, timings_ms, acc_out128);
  });
}

extern "C" int hd_aggregate_end_to_end(int mos, int tkind, const uint8_t* protocol, size_t plen, const uint8_t* instances,
                                       size_t ilen, const uint8_t* proofs, size_t prlen, uint32_t n, const uint8_t* dk320,
                                       unsigned threads, double* timings_ms, uint8_t* acc_out128) {
  return mos == 0 ? aggregate_impl<Gwc19>(tkind, protocol, plen, instances, ilen, proofs, prlen, n, dk320, threads,
                                          timings_ms, acc_out128)
                  : aggregate_impl<Bdfg21>(tkind, protocol, plen, instances, ilen, proofs, prlen, n, dk320, threads,
                                           timings_ms, acc_out128);
}

// `PlonkSuccinctVerifier` only (no pairing): the per-rank step of proof-sharded aggregation
extern "C" int hd_plonk_succinct_verify(int mos, int tkind, const uint8_t* protocol, size_t plen, const uint8_t* instances,
                                        size_t ilen, const uint8_t* proofs, size_t prlen, uint32_t n,
                                        const uint8_t* dk320, uint8_t* accs_out, size_t accs_cap, uint32_t* n_accs) {
  return mos == 0 ? plonk_verify_impl<Gwc19>(tkind, protocol, plen, instances, ilen, proofs, prlen, n, dk320, accs_out,
                                             accs_cap, n_accs, false)
                  : plonk_verify_impl<Bdfg21>(tkind, protocol, plen, instances, ilen, proofs, prlen, n, dk320, accs_out,
                                              accs_cap, n_accs, false);
}

// dev aid (CPU only, one thread): microseconds per proof of the host half's phases for Gwc19 + Keccak transcripts --
// out[0..6] = read_proof, CommonPolyEval, evaluations_map, commitments, queries, pcs_msms, pairs.  tools/host_phases.py
extern "C" int hd_plonk_host_phases(const uint8_t* protocol, size_t plen, const uint8_t* instances, size_t ilen,
                                    const uint8_t* proofs, size_t prlen, uint32_t n, const uint8_t* dk320, int reps, double* out7) {
  return guarded([&] {
    using clk = std::chrono::steady_clock;
    PlonkProtocol pr = parse_protocol(protocol, plen);
    KzgDecidingKey dk(G1Affine::from_bytes(dk320), G2Affine::from_bytes(dk320 + 64), G2Affine::from_bytes(dk320 + 192));
    std::vector<std::vector<std::vector<Fr>>> insts;
    std::vector<std::vector<uint8_t>> pbytes;
    wire::split_batch(instances, ilen, proofs, prlen, n, insts, pbytes);
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    auto lap = [&](clk::time_point& t, int k) {
      auto now = clk::now();
      acc[k] += std::chrono::duration<double, std::micro>(now - t).count();
      t = now;
    };
    size_t sink = 0;
    for (int r = 0; r < reps; ++r)
      for (uint32_t i = 0; i < n; ++i) {
        auto t = clk::now();
        EvmTranscript tr(pbytes[i]);
        auto pf = PlonkSuccinctVerifier<Gwc19>::read_proof(dk.svk, pr, insts[i], tr);
        if (!pf.ok()) return error_code(pf.err);
        const auto& proof = *pf.value;
        lap(t, 0);
        CommonPolyEval cpe(pr.domain, pr.langranges(), proof.z);
        lap(t, 1);
        auto evals = proof.evaluations_map(pr, insts[i], cpe);
        lap(t, 2);
        auto cm = proof.commitments(pr, cpe, evals);
        lap(t, 3);
        auto queries = proof.queries(pr, evals);
        lap(t, 4);
        auto [lhs, rhs] = plonk_detail::pcs_msms(cm, proof.z, queries, proof.pcs);
        lap(t, 5);
        auto a = lhs.pairs(dk.svk.g), b = rhs.pairs(dk.svk.g);
        sink += a.size() + b.size();
        lap(t, 6);
      }
    for (int k = 0; k < 7; ++k) out7[k] = acc[k] / ((double)reps * n);
    return sink ? 0 : -1;
  });
}

// KzgAs::create_proof (non-zk, fresh Keccak transcript) over m accumulators, then decide: the combine
// step every rank runs on the gathered accumulators.  Returns 1 / 0; acc_out128 = the folded accumulator.
extern "C" int hd_kzg_as_accumulate_and_decide(const uint8_t* accs128, uint32_t m, const uint8_t* dk320,
                                               uint8_t* acc_out128) {
  return guarded([&] {
    KzgDecidingKey dk(G1Affine::from_bytes(dk320), G2Affine::from_bytes(dk320 + 64), G2Affine::from_bytes(dk320 + 192));
    std::vector<KzgAccumulator> accs;
    for (uint32_t i = 0; i < m; ++i)
      accs.push_back(KzgAccumulator{G1Affine::from_bytes(accs128 + 128 * i), G1Affine::from_bytes(accs128 + 128 * i + 64)});
    EvmTranscript t;
    auto acc = KzgAs<Gwc19>::create_proof(KzgAsProvingKey{}, accs, t, Fr());
    if (!acc.ok()) return error_code(acc.err);
    if (acc_out128) acc.value->to_bytes(acc_out128);
    return KzgAs<Gwc19>::decide(dk, *acc.value).ok() ? 1 : 0;
  });
}

// `CostEstimation` of PlonkVerifier<MOS> (verifier/plonk.rs:149-188): out[0..4] = instance, commitment, evaluation, msm, pairing
extern "C" int hd_plonk_estimate_cost(int mos, const uint8_t* protocol, size_t plen, uint64_t* out5) {
  return guarded([&] {
    PlonkProtocol pr = parse_protocol(protocol, plen);
    Cost c = mos == 0 ? PlonkVerifier<Gwc19>::estimate_cost(pr) : PlonkVerifier<Bdfg21>::estimate_cost(pr);
    out5[0] = c.num_instance;
    out5[1] = c.num_commitment;
    out5[2] = c.num_evaluation;
    out5[3] = c.num_msm;
    out5[4] = c.num_pairing;
    return 0;
  });
}

extern "C" int hd_plonk_verify(int mos, int tkind, const uint8_t* protocol, size_t plen, const uint8_t* instances,
                               size_t ilen, const uint8_t* proofs, size_t prlen, uint32_t n, const uint8_t* dk320,
                               uint8_t* accs_out, size_t accs_cap, uint32_t* n_accs) {
  return mos == 0 ? plonk_verify_impl<Gwc19>(tkind, protocol, plen, instances, ilen, proofs, prlen, n, dk320, accs_out,
                                             accs_cap, n_accs)
                  : plonk_verify_impl<Bdfg21>(tkind, protocol, plen, instances, ilen, proofs, prlen, n, dk320, accs_out,
                                              accs_cap, n_accs);
}


// ---- IPA (pcs/ipa.rs, pcs/ipa/{accumulation,decider}.rs): entry points in ipa_driver.inc -------------
#include "ipa.hpp"
namespace {
std::unique_ptr<Transcript> make_transcript(int tkind, const uint8_t* proof, size_t plen) {
  std::vector<uint8_t> bytes(proof, proof + plen);
  if (tkind == 0) return std::make_unique<EvmTranscript>(std::move(bytes));
  return std::make_unique<PoseidonTranscript>(std::move(bytes));
}
}  // namespace
#define SNARKV_DRV(name) hd_##name
#include "ipa_driver.inc"
