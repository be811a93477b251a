/* bn254_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped,
 * never on the product path; see oracle/README.md).
 *
 * C restatement of the reference's CPU algorithms for the KZG hot path, over a
 * 4 x 64-bit Montgomery Fq (R = 2^256) and Jacobian G1 -- the representation
 * halo2curves 0.6.0 uses (reference snark-verifier/Cargo.toml:14; the crate is
 * not vendored and there is no Rust toolchain here, so this is a restatement
 * from the mathematical definitions: PARITY UNPINNED by reference data, pinned
 * by the public constants / algebraic invariants in tests/ and by agreement
 * with the independent big-integer oracle oracle/bn254.py).
 *
 *   oracle_g1_msm_naive      <- NativeLoader::multi_scalar_multiplication,
 *                               reference snark-verifier/src/loader/native.rs:61-71
 *   oracle_g1_msm_pippenger  <- util::msm::multi_scalar_multiplication(+_serial),
 *                               reference snark-verifier/src/util/msm.rs:259-343
 *                               (same window rule, unsigned digits, bucket
 *                               running sum, chunk-per-thread split :311-336)
 *
 * "reference algorithm restated in C -- not a halo2curves measurement".
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fq;
typedef struct { fq x, y, z; } g1j;       /* Jacobian; identity <=> z == 0 */
typedef struct { fq x, y; } g1a;          /* affine Montgomery; identity = (0,0) */

static const uint64_t P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull,
                              0x30644e72e131a029ull};
static const uint64_t PINV = 0x87d20782e4866389ull; /* -p^-1 mod 2^64 */
static const fq R2 = {{0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull,
                       0x06d89f71cab8351full}};
static const fq ONE = {{0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull,
                        0x0e0a77c19a07df2full}};
static const uint64_t RMOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull,
                                 0x30644e72e131a029ull};

static inline int fq_is_zero(const fq* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int fq_eq(const fq* a, const fq* b) {
  return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}

static inline void reduce_once(uint64_t t[4]) {
  uint64_t d[4];
  u128 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 x = (u128)t[i] - P[i] - (uint64_t)br;
    d[i] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
  if (!br) memcpy(t, d, 32);
}

static inline void fq_add(fq* r, const fq* a, const fq* b) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a->v[i] + b->v[i];
    r->v[i] = (uint64_t)c;
    c >>= 64;
  }
  reduce_once(r->v);
}

static inline void fq_sub(fq* r, const fq* a, const fq* b) {
  u128 br = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; ++i) {
    u128 x = (u128)a->v[i] - b->v[i] - (uint64_t)br;
    t[i] = (uint64_t)x;
    br = (x >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (u128)t[i] + P[i];
      t[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  memcpy(r->v, t, 32);
}

static inline void fq_mul(fq* r, const fq* a, const fq* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a->v[i] * b->v[j] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * PINV;
    c = ((u128)m * P[0] + t[0]) >> 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * P[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  reduce_once(t);
  memcpy(r->v, t, 32);
}

static inline void fq_sqr(fq* r, const fq* a) { fq_mul(r, a, a); }
static inline void fq_dbl(fq* r, const fq* a) { fq_add(r, a, a); }

static void fq_pow(fq* r, const fq* a, const uint64_t e[4]) {
  fq res = ONE;
  for (int i = 3; i >= 0; --i)
    for (int b = 63; b >= 0; --b) {
      fq_sqr(&res, &res);
      if ((e[i] >> b) & 1) fq_mul(&res, &res, a);
    }
  *r = res;
}

static void fq_inv(fq* r, const fq* a) {
  uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
  fq_pow(r, a, e);
}

static void fq_from_bytes(fq* r, const uint8_t* b) {
  fq t;
  memcpy(t.v, b, 32); /* little-endian host */
  fq_mul(r, &t, &R2);
}

static void fq_to_bytes(uint8_t* b, const fq* a) {
  fq one = {{1, 0, 0, 0}}, t;
  fq_mul(&t, a, &one);
  memcpy(b, t.v, 32);
}

/* ---------------------------------------------------------------- G1 */
static void g1j_identity(g1j* r) { memset(r, 0, sizeof(*r)); }
static int g1j_is_identity(const g1j* p) { return fq_is_zero(&p->z); }
static int g1a_is_identity(const g1a* p) { return fq_is_zero(&p->x) && fq_is_zero(&p->y); }

static void g1j_double(g1j* r, const g1j* p) { /* dbl-2009-l, a = 0 */
  if (g1j_is_identity(p)) { g1j_identity(r); return; }
  fq a, b, c, d, e, f, t, x3, y3, z3;
  fq_sqr(&a, &p->x);
  fq_sqr(&b, &p->y);
  fq_sqr(&c, &b);
  fq_add(&t, &p->x, &b);
  fq_sqr(&t, &t);
  fq_sub(&t, &t, &a);
  fq_sub(&t, &t, &c);
  fq_dbl(&d, &t);
  fq_dbl(&e, &a);
  fq_add(&e, &e, &a);
  fq_sqr(&f, &e);
  fq_dbl(&t, &d);
  fq_sub(&x3, &f, &t);
  fq_mul(&z3, &p->y, &p->z);
  fq_dbl(&z3, &z3);
  fq_sub(&t, &d, &x3);
  fq_mul(&y3, &e, &t);
  fq_dbl(&c, &c);
  fq_dbl(&c, &c);
  fq_dbl(&c, &c);
  fq_sub(&y3, &y3, &c);
  r->x = x3; r->y = y3; r->z = z3;
}

static void g1j_add_mixed(g1j* r, const g1j* p, const g1a* q) { /* madd-2007-bl */
  if (g1a_is_identity(q)) { *r = *p; return; }
  if (g1j_is_identity(p)) { r->x = q->x; r->y = q->y; r->z = ONE; return; }
  fq z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
  fq_sqr(&z1z1, &p->z);
  fq_mul(&u2, &q->x, &z1z1);
  fq_mul(&s2, &q->y, &p->z);
  fq_mul(&s2, &s2, &z1z1);
  fq_sub(&h, &u2, &p->x);
  fq_sub(&rr, &s2, &p->y);
  if (fq_is_zero(&h)) {
    if (fq_is_zero(&rr)) { g1j_double(r, p); return; }
    g1j_identity(r);
    return;
  }
  fq_sqr(&hh, &h);
  fq_dbl(&i, &hh);
  fq_dbl(&i, &i);
  fq_mul(&j, &h, &i);
  fq_dbl(&rr, &rr);
  fq_mul(&v, &p->x, &i);
  fq_sqr(&x3, &rr);
  fq_sub(&x3, &x3, &j);
  fq_sub(&x3, &x3, &v);
  fq_sub(&x3, &x3, &v);
  fq_sub(&t, &v, &x3);
  fq_mul(&y3, &rr, &t);
  fq_mul(&t, &p->y, &j);
  fq_dbl(&t, &t);
  fq_sub(&y3, &y3, &t);
  fq_add(&z3, &p->z, &h);
  fq_sqr(&z3, &z3);
  fq_sub(&z3, &z3, &z1z1);
  fq_sub(&z3, &z3, &hh);
  r->x = x3; r->y = y3; r->z = z3;
}

static void g1j_add(g1j* r, const g1j* p, const g1j* q) { /* add-2007-bl */
  if (g1j_is_identity(q)) { *r = *p; return; }
  if (g1j_is_identity(p)) { *r = *q; return; }
  fq z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
  fq_sqr(&z1z1, &p->z);
  fq_sqr(&z2z2, &q->z);
  fq_mul(&u1, &p->x, &z2z2);
  fq_mul(&u2, &q->x, &z1z1);
  fq_mul(&s1, &p->y, &q->z);
  fq_mul(&s1, &s1, &z2z2);
  fq_mul(&s2, &q->y, &p->z);
  fq_mul(&s2, &s2, &z1z1);
  fq_sub(&h, &u2, &u1);
  fq_sub(&rr, &s2, &s1);
  if (fq_is_zero(&h)) {
    if (fq_is_zero(&rr)) { g1j_double(r, p); return; }
    g1j_identity(r);
    return;
  }
  fq_dbl(&i, &h);
  fq_sqr(&i, &i);
  fq_mul(&j, &h, &i);
  fq_dbl(&rr, &rr);
  fq_mul(&v, &u1, &i);
  fq_sqr(&x3, &rr);
  fq_sub(&x3, &x3, &j);
  fq_sub(&x3, &x3, &v);
  fq_sub(&x3, &x3, &v);
  fq_sub(&t, &v, &x3);
  fq_mul(&y3, &rr, &t);
  fq_mul(&t, &s1, &j);
  fq_dbl(&t, &t);
  fq_sub(&y3, &y3, &t);
  fq_add(&z3, &p->z, &q->z);
  fq_sqr(&z3, &z3);
  fq_sub(&z3, &z3, &z1z1);
  fq_sub(&z3, &z3, &z2z2);
  fq_mul(&z3, &z3, &h);
  r->x = x3; r->y = y3; r->z = z3;
}

static void g1j_to_affine(g1a* r, const g1j* p) {
  if (g1j_is_identity(p)) { memset(r, 0, sizeof(*r)); return; }
  fq zi, zi2, zi3;
  fq_inv(&zi, &p->z);
  fq_sqr(&zi2, &zi);
  fq_mul(&zi3, &zi2, &zi);
  fq_mul(&r->x, &p->x, &zi2);
  fq_mul(&r->y, &p->y, &zi3);
}

static void g1a_from_bytes(g1a* r, const uint8_t* b) {
  fq_from_bytes(&r->x, b);
  fq_from_bytes(&r->y, b + 32);
}
static void g1a_to_bytes(uint8_t* b, const g1a* p) {
  fq_to_bytes(b, &p->x);
  fq_to_bytes(b + 32, &p->y);
}

/* base * scalar: 256-step left-to-right double-and-add (native.rs:67) */
static void g1_scalar_mul(g1j* r, const g1a* base, const uint8_t k[32]) {
  g1j acc;
  g1j_identity(&acc);
  for (int i = 255; i >= 0; --i) {
    g1j_double(&acc, &acc);
    if ((k[i >> 3] >> (i & 7)) & 1) g1j_add_mixed(&acc, &acc, base);
  }
  *r = acc;
}

/* ------------------------------------------------- native.rs:61-71 */
int oracle_g1_msm_naive(const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t out[64]) {
  if (n == 0) return -1; /* reference: reduce().unwrap() panics */
  g1j acc;
  g1j_identity(&acc);
  for (size_t i = 0; i < n; ++i) {
    g1a b;
    g1j t;
    g1a_from_bytes(&b, points + 64 * i);
    g1_scalar_mul(&t, &b, scalars + 32 * i);
    g1j_add(&acc, &acc, &t);
  }
  g1a a;
  g1j_to_affine(&a, &acc);
  g1a_to_bytes(out, &a);
  return 0;
}

/* segmented form (same per-segment semantics) */
int oracle_g1_msm_batched(const uint8_t* scalars, const uint8_t* points, const uint32_t* offsets, size_t n_msm,
                          uint8_t* out) {
  for (size_t k = 0; k < n_msm; ++k) {
    int rc = oracle_g1_msm_naive(scalars + 32 * (size_t)offsets[k], points + 64 * (size_t)offsets[k],
                                 offsets[k + 1] - offsets[k], out + 64 * k);
    if (rc) return rc;
  }
  return 0;
}

/* ------------------------------------------------- msm.rs:229-304 */
typedef struct { int tag; g1a aff; g1j proj; } bucket_t; /* enum Bucket { None, Affine, Projective } :229-233 */

static void msm_serial(const uint8_t* scalars, const g1a* bases, size_t n, g1j* result) {
  const int num_bits = 256; /* 8 * repr length, msm.rs:265-266 */
  int c = (int)ceil(log((double)n)) + 2; /* msm.rs:268 */
  size_t num_buckets = ((size_t)1 << c) - 1;
  int num_window = (num_bits + c - 1) / c;
  bucket_t* buckets = (bucket_t*)malloc(num_buckets * sizeof(bucket_t));
  for (int idx = num_window - 1; idx >= 0; --idx) {
    for (int k = 0; k < c; ++k) g1j_double(result, result); /* msm.rs:285-287 */
    for (size_t b = 0; b < num_buckets; ++b) buckets[b].tag = 0;
    for (size_t i = 0; i < n; ++i) {
      /* windowed_scalar, msm.rs:271-281: 8-byte LE load at skip_bytes */
      size_t skip_bits = (size_t)idx * c, skip_bytes = skip_bits / 8;
      uint8_t v8[8] = {0};
      for (size_t k = 0; k < 8 && skip_bytes + k < 32; ++k) v8[k] = scalars[32 * i + skip_bytes + k];
      uint64_t v;
      memcpy(&v, v8, 8);
      size_t d = (size_t)((v >> (skip_bits - skip_bytes * 8)) & num_buckets);
      if (d != 0) { /* msm.rs:293 */
        bucket_t* bk = &buckets[d - 1];
        if (bk->tag == 0) { bk->tag = 1; bk->aff = bases[i]; }
        else if (bk->tag == 1) {
          g1j t;
          if (g1a_is_identity(&bk->aff)) g1j_identity(&t);
          else { t.x = bk->aff.x; t.y = bk->aff.y; t.z = ONE; }
          g1j_add_mixed(&bk->proj, &t, &bases[i]);
          bk->tag = 2;
        } else g1j_add_mixed(&bk->proj, &bk->proj, &bases[i]);
      }
    }
    g1j running;
    g1j_identity(&running);
    for (size_t b = num_buckets; b-- > 0;) { /* msm.rs:298-302 */
      if (buckets[b].tag == 1) g1j_add_mixed(&running, &running, &buckets[b].aff);
      else if (buckets[b].tag == 2) g1j_add(&running, &buckets[b].proj, &running);
      g1j_add(result, result, &running);
    }
  }
  free(buckets);
}

typedef struct { const uint8_t* scalars; const g1a* bases; size_t n; g1j result; } chunk_job;
static void* chunk_main(void* arg) {
  chunk_job* j = (chunk_job*)arg;
  g1j_identity(&j->result);
  msm_serial(j->scalars, j->bases, j->n, &j->result);
  return NULL;
}

/* msm.rs:308-343; `threads` plays rayon's current_num_threads() (1 = the
 * non-"parallel" build).  Output: to_affine() of the projective result. */
int oracle_g1_msm_pippenger(const uint8_t* scalars, const uint8_t* points, size_t n, int threads, uint8_t out[64]) {
  if (n == 0) return -1; /* reference indexes scalars[0], msm.rs:265 */
  g1a* bases = (g1a*)malloc(n * sizeof(g1a));
  for (size_t i = 0; i < n; ++i) g1a_from_bytes(&bases[i], points + 64 * i);
  g1j total;
  g1j_identity(&total);
  if (threads <= 1 || n < (size_t)threads) { /* msm.rs:316-320 */
    msm_serial(scalars, bases, n, &total);
  } else {
    size_t chunk = (n + threads - 1) / threads; /* msm.rs:322 */
    size_t njobs = (n + chunk - 1) / chunk;
    chunk_job* jobs = (chunk_job*)calloc(njobs, sizeof(chunk_job));
    pthread_t* th = (pthread_t*)calloc(njobs, sizeof(pthread_t));
    for (size_t k = 0; k < njobs; ++k) {
      size_t lo = k * chunk, hi = lo + chunk < n ? lo + chunk : n;
      jobs[k].scalars = scalars + 32 * lo;
      jobs[k].bases = bases + lo;
      jobs[k].n = hi - lo;
      pthread_create(&th[k], NULL, chunk_main, &jobs[k]);
    }
    for (size_t k = 0; k < njobs; ++k) {
      pthread_join(th[k], NULL);
      g1j_add(&total, &total, &jobs[k].result); /* msm.rs:333-335 */
    }
    free(jobs);
    free(th);
  }
  free(bases);
  g1a a;
  g1j_to_affine(&a, &total);
  g1a_to_bytes(out, &a);
  return 0;
}

/* ---------------------------------------------------- small helpers */
int oracle_g1_add(const uint8_t* p, const uint8_t* q, uint8_t out[64]) {
  g1a a, b, r;
  g1j t;
  g1a_from_bytes(&a, p);
  g1a_from_bytes(&b, q);
  if (g1a_is_identity(&a)) g1j_identity(&t);
  else { t.x = a.x; t.y = a.y; t.z = ONE; }
  g1j_add_mixed(&t, &t, &b);
  g1j_to_affine(&r, &t);
  g1a_to_bytes(out, &r);
  return 0;
}

int oracle_g1_mul(const uint8_t* p, const uint8_t* k, uint8_t out[64]) {
  g1a a, r;
  g1j t;
  g1a_from_bytes(&a, p);
  g1_scalar_mul(&t, &a, k);
  g1j_to_affine(&r, &t);
  g1a_to_bytes(out, &r);
  return 0;
}

int oracle_g1_is_on_curve(const uint8_t* p) {
  g1a a;
  g1a_from_bytes(&a, p);
  if (g1a_is_identity(&a)) return 1;
  fq l, r, three, t;
  fq_sqr(&l, &a.y);
  fq_sqr(&r, &a.x);
  fq_mul(&r, &r, &a.x);
  fq_add(&t, &ONE, &ONE);
  fq_add(&three, &t, &ONE);
  fq_add(&r, &r, &three);
  return fq_eq(&l, &r);
}

/* ------------------------------------------- synthetic input sampling
 * SplitMix64 streams.  Scalars: 4 words, top two bits cleared
 * (< 2^254), one conditional subtraction of r.  Points: x from the stream,
 * incremented until x^3+3 is a square (p = 3 mod 4: y = (x^3+3)^((p+1)/4)),
 * the root with even canonical y.  G1 has cofactor 1, so every curve point is
 * in the group. */
static uint64_t splitmix64(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void oracle_sample_scalars(uint64_t seed, size_t first, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; ++i) {
    uint64_t s = seed ^ (0xA5A5A5A500000000ull + (uint64_t)(first + i) * 0x9E3779B97F4A7C15ull);
    uint64_t w[4];
    for (int j = 0; j < 4; ++j) w[j] = splitmix64(&s);
    w[3] &= 0x3FFFFFFFFFFFFFFFull;
    uint64_t d[4];
    u128 br = 0;
    for (int j = 0; j < 4; ++j) {
      u128 x = (u128)w[j] - RMOD[j] - (uint64_t)br;
      d[j] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
    memcpy(out + 32 * i, br ? w : d, 32);
  }
}

void oracle_sample_points(uint64_t seed, size_t first, size_t n, uint8_t* out) {
  static const uint64_t EXP[4] = {0x4f082305b61f3f52ull, 0x65e05aa45a1c72a3ull, 0x6e14116da0605617ull,
                                  0x0c19139cb84c680aull}; /* (p+1)/4 */
  fq three, t;
  fq_add(&t, &ONE, &ONE);
  fq_add(&three, &t, &ONE);
  for (size_t i = 0; i < n; ++i) {
    uint64_t s = seed ^ (0x5A5A5A5A00000000ull + (uint64_t)(first + i) * 0x9E3779B97F4A7C15ull);
    uint64_t w[4];
    for (int j = 0; j < 4; ++j) w[j] = splitmix64(&s);
    w[3] &= 0x0FFFFFFFFFFFFFFFull; /* < 2^252 < p */
    fq xm, rhs, y, y2;
    fq xc;
    memcpy(xc.v, w, 32);
    for (;;) {
      fq_mul(&xm, &xc, &R2);
      fq_sqr(&rhs, &xm);
      fq_mul(&rhs, &rhs, &xm);
      fq_add(&rhs, &rhs, &three);
      fq_pow(&y, &rhs, EXP);
      fq_sqr(&y2, &y);
      if (fq_eq(&y2, &rhs)) break;
      xc.v[0] += 1; /* no carry concern: probability 2^-64 */
    }
    uint8_t yb[32];
    fq_to_bytes(yb, &y);
    if (yb[0] & 1) {
      fq z = {{0, 0, 0, 0}};
      fq_sub(&y, &z, &y);
      fq_to_bytes(yb, &y);
    }
    fq_to_bytes(out + 64 * i, &xm);
    memcpy(out + 64 * i + 32, yb, 32);
  }
}

/* ------------------------------------------- pairing side (KzgAs::decide, decider.rs:70-93) */
#include "bn254_pairing.inc"
