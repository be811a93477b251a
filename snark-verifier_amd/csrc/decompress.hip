// Batched G1 point decompression on the device.
//
// The reference's native Poseidon transcript reads every commitment of a proof as a COMPRESSED point
// (`C::from_bytes(&data)`, snark-verifier/src/system/halo2/transcript/halo2.rs:260-273; the curve code is
// halo2curves', Cargo.toml:14): one square root in Fq -- ~380 field products -- per point, ~13 per proof.  On the
// host that is 0.15 ms per proof, 2.4 ms of a 1 024-proof aggregation on 64 threads and more than the hashing it
// precedes.  The points of a batch are independent and the exponent (p + 1) / 4 is the same for all of them, so
// the whole batch is one launch of lock-step ladders on the lazy 9 x 29-bit field:
//
//   k_g1_decompress : one lane per point.  32 bytes in = x little-endian canonical, bit 254 = parity of y,
//                     bit 255 = identity (everything else zero);  64 bytes out = x || y canonical (the
//                     identity: 64 zero bytes), ok = 0 for an invalid encoding (x >= p, x^3 + 3 not a
//                     square, a malformed identity) -- the cases `from_bytes` answers `None` to.
//
// The encoding is halo2curves 0.6.0 bn256's as the host mirror has it (host/transcript.hpp `g1_decompress`, the
// function this accelerates and the one its callers fall back to); it is a crate internal, recalled, not pinned
// by reference data.
#include "ctx.hpp"
#include "fq.h"
#include "fq29.h"

namespace snarkv {

// a^((p + 1) / 4): the square root of a square, for p = 3 (mod 4).  Lane-uniform exponent: no divergence.
__device__ __noinline__ Fq29 fq29_pow_p_plus_1_over_4(const Fq29 a) {  // by value: registers, not the stack
  constexpr uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u,
                             0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};  // (p + 1) / 4 < 2^253
  Fq29 res = fq29_one();
  for (int i = 7; i >= 0; --i) {
    const uint32_t w = e[i];
    for (int b = 31; b >= 0; --b) {
      res = fq29_sqr(fq29_norm(res));
      if ((w >> b) & 1u) res = fq29_mul(res, a);
    }
  }
  return res;
}

// `offs` == nullptr: encoding i at in + 8 i words.  Otherwise the encodings lie inside records of `stride_words` words
// (proofs): encoding i = point i % P of record i / P, at word offs[i % P] of it (16-byte aligned).
__global__ void __launch_bounds__(64) k_g1_decompress(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                       uint8_t* __restrict__ ok, uint32_t n, uint32_t mont,
                                                       const uint32_t* __restrict__ offs, uint32_t P, uint32_t stride_words) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* src = reinterpret_cast<const uint4*>(offs ? in + (size_t)(i / P) * stride_words + offs[i % P] : in + (size_t)i * 8);
  const uint4 a = src[0], b = src[1];
  uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  const uint32_t is_inf = w[7] >> 31, ysign = (w[7] >> 30) & 1u;
  w[7] &= 0x3FFFFFFFu;
  uint32_t o[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) o[j] = 0u;
  bool good = fq_canonical_in_range(w);  // x < p
  if (good && is_inf) {
    // the identity has exactly one encoding: the flag and nothing else
    uint32_t any = ysign;
#pragma unroll
    for (int j = 0; j < 8; ++j) any |= w[j];
    good = any == 0u;
  } else if (good) {
    const Fq29 x = fq29_from_canonical(w);  // Montgomery, product output
    uint32_t three[8] = {3u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const Fq29 y2 = fq29_norm(fq29_add(fq29_mul(fq29_sqr(x), x), fq29_from_canonical(three)));  // x^3 + 3
    const Fq29 y = fq29_pow_p_plus_1_over_4(y2);
    uint32_t chk[8], want[8], yc[8];
    fq29_to_canonical(fq29_sqr(fq29_norm(y)), chk);
    fq29_to_canonical(y2, want);
    uint32_t diff = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) diff |= chk[j] ^ want[j];
    good = diff == 0u;  // otherwise x^3 + 3 is not a square: no such point
    fq29_to_canonical(y, yc);
    if ((yc[0] & 1u) != ysign) {
      // the other root, p - y (y = 0 would have no other root; x^3 + 3 = 0 has no solution with a parity flag to honour,
      // and the host function this mirrors leaves 0 as it is)
      uint32_t nz = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) nz |= yc[j];
      if (nz != 0u) fq29_to_canonical(fq29_norm(fq29_neg(fq29_norm(y))), yc);
    }
    if (good) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = w[j];
        o[8 + j] = yc[j];
      }
      if (mont) {  // SNARKV_FLAG_MONTGOMERY: the point in halo2curves' in-memory form (the compressed input is the wire form)
        fq29_to_words(x, o, true);
        fq29_to_words(fq29_from_canonical(yc), o + 8, true);
      }
    }
  }
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)i * 16);
#pragma unroll
  for (int j = 0; j < 4; ++j) dst[j] = make_uint4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
  ok[i] = good ? 1 : 0;
}

int launch_g1_decompress(snarkv_ctx* ctx, const void* d_in32, size_t n, void* d_out64, void* d_ok) {
  hipLaunchKernelGGL(k_g1_decompress, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint32_t*)d_in32,
                     (uint32_t*)d_out64, (uint8_t*)d_ok, (uint32_t)n, ctx->mont ? 1u : 0u, (const uint32_t*)nullptr, 1u, 0u);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

// the P points of each of n_rec records (canonical output whatever the context's flags: these feed a transcript)
int launch_g1_decompress_records(snarkv_ctx* ctx, const void* d_records, size_t n_rec, size_t stride_words, const void* d_offs_words,
                                 size_t P, void* d_out64, void* d_ok) {
  const size_t n = n_rec * P;
  hipLaunchKernelGGL(k_g1_decompress, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint32_t*)d_records,
                     (uint32_t*)d_out64, (uint8_t*)d_ok, (uint32_t)n, 0u, (const uint32_t*)d_offs_words, (uint32_t)P,
                     (uint32_t)stride_words);
  SNARKV_HIP(hipGetLastError());
  return SNARKV_OK;
}

}  // namespace snarkv
