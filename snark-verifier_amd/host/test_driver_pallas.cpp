// Test driver of the PASTA flavour of the host mirror -> libsnarkv_hosttest_pallas.so (compiled with
// -DSNARKV_HOST_PALLAS: `Fr` = pallas::Scalar, the loader bound to libsnarkv_pallas.so).  It carries the
// curve-generic part of the mirror -- Msm, the native loader, the IPA layer (host/ipa.hpp), the PLONK
// verifier over `IpaAs<Bgh19>` (host/plonk.hpp) -- with halo2's
// Blake2b transcript: the reference's own `test_ipa` / `test_ipa_as` setting (pcs/ipa.rs:434-466,
// pcs/ipa/accumulation.rs:240-290, system/halo2/test/ipa/native.rs).  KZG (pairing) and the EVM / Poseidon
// transcripts stay BN254.
#ifndef SNARKV_HOST_PALLAS
#error "compile with -DSNARKV_HOST_PALLAS"
#endif
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>

#include "blake2b_transcript.hpp"
#include "ipa.hpp"
#include "plonk.hpp"
#include "wire.hpp"

using namespace snarkv_host;

namespace {
#include "driver_parse.inc"

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
std::unique_ptr<Transcript> make_transcript(int /*tkind: Blake2b only*/, const uint8_t* proof, size_t plen) {
  return std::make_unique<Blake2bTranscript>(std::vector<uint8_t>(proof, proof + plen));
}
}  // namespace

#define SNARKV_DRV(name) hp_##name
#include "ipa_driver.inc"

extern "C" {
// Fr = pallas::Scalar self-test hooks
void hp_fr_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fr x, y;
  Fr::from_bytes(a, &x);
  Fr::from_bytes(b, &y);
  (x * y).to_bytes(out);
}
int hp_fr_inv(const uint8_t* a, uint8_t* out) {
  Fr x, y;
  Fr::from_bytes(a, &x);
  if (!x.invert(&y)) return 0;
  y.to_bytes(out);
  return 1;
}
// BLAKE2b-512 with a 16-byte personalisation (RFC 7693), fed in `chunk`-byte pieces
void hp_blake2b(const char* person16, const uint8_t* data, size_t len, size_t chunk, uint8_t out[64]) {
  Blake2b h(64, person16);
  for (size_t o = 0; o < len; o += chunk) h.update(data + o, std::min(chunk, len - o));
  h.digest(out);
}
// a scripted transcript session: ops = sequence of bytes: 'P' read point, 'S' read scalar, 'C' squeeze;
// out = for each op the value (64 / 32 / 32 bytes).  Returns the number of ops done, or -10 at the first
// Transcript error.
int hp_transcript_script(const uint8_t* proof, size_t plen, const char* ops, size_t n_ops, uint8_t* out) {
  return guarded([&] {
    Blake2bTranscript t(std::vector<uint8_t>(proof, proof + plen));
    size_t o = 0;
    for (size_t i = 0; i < n_ops; ++i) {
      if (ops[i] == 'P') {
        auto p = t.read_ec_point();
        if (!p.ok()) return -10;
        memcpy(out + o, p.value->b, 64);
        o += 64;
      } else if (ops[i] == 'S') {
        auto s = t.read_scalar();
        if (!s.ok()) return -10;
        s.value->to_bytes(out + o);
        o += 32;
      } else {
        t.squeeze_challenge().to_bytes(out + o);
        o += 32;
      }
    }
    return (int)n_ops;
  });
}
}
