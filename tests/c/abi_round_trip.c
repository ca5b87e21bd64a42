/* A plain C consumer of include/ntru_engine.h: what a C/C++ host (or a cgo / JNI / N-API stub) sees of the engine.
 * No HIP header, no Python: host buffers in, host buffers out, every witness array compared bit for bit with the CPU
 * oracle (TEST INFRASTRUCTURE, linked here as the checker only).  Built and run by tests/test_c_abi_gpu.py:
 *   gcc -O2 -std=c11 -Iinclude tests/c/abi_round_trip.c -o <tmp>/abi_round_trip -L<lib> -lntru_engine -L oracle -lntru_oracle
 * Exit code 0 = identical; prints one line per parameter set. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ntru_engine.h"

/* oracle/ntru_oracle.c (encryptBits index.js:87-110, decryptBits index.js:111-140); mode 0 = exact integer arithmetic */
int orc_encrypt_batch(int N, int q, const uint16_t *h, const uint8_t *r, const uint8_t *m, int64_t B, uint16_t *e,
                      uint16_t *quotE, int mode);
int orc_decrypt_batch(int N, int q, int p, const int8_t *f, const uint8_t *fp, const uint16_t *e, int64_t B,
                      uint8_t *value, uint16_t *quot1, uint16_t *rem1, uint8_t *quot2, int mode);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {                      /* xorshift64*: any fixed stream will do */
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}

#define CHECK(call)                                                                       \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ != NTRU_OK) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ntru_last_error());                   \
      return 2;                                                                           \
    }                                                                                     \
  } while (0)

static int one_set(ntru_engine_t *eng, int N, int q, int p, int64_t B) {
  const size_t n = (size_t)B * N;
  uint16_t *h = malloc(2 * (size_t)N), *e = malloc(2 * n), *qe = malloc(2 * n), *q1 = malloc(2 * n), *r1 = malloc(2 * n);
  uint16_t *e_o = malloc(2 * n), *qe_o = malloc(2 * n), *q1_o = malloc(2 * n), *r1_o = malloc(2 * n);
  int8_t *f = malloc(N);
  uint8_t *fp = malloc(N), *r = malloc(n), *m = malloc(n), *v = malloc(n), *q2 = malloc(n), *v_o = malloc(n), *q2_o = malloc(n);
  for (int i = 0; i < N; i++) { h[i] = (uint16_t)(rnd() % (uint32_t)q); f[i] = (int8_t)(rnd() % 3) - 1; fp[i] = (uint8_t)(rnd() % (uint32_t)p); }
  for (size_t i = 0; i < n; i++) { r[i] = (uint8_t)(rnd() % 3); m[i] = (uint8_t)(rnd() & 1); }
  CHECK(ntru_encrypt_batch(eng, N, q, h, r, m, B, e, qe));
  CHECK(ntru_decrypt_batch(eng, N, q, p, f, fp, e, B, v, q1, r1, q2));
  if (orc_encrypt_batch(N, q, h, r, m, B, e_o, qe_o, 0) || orc_decrypt_batch(N, q, p, f, fp, e_o, B, v_o, q1_o, r1_o, q2_o, 0)) {
    fprintf(stderr, "oracle failed\n");
    return 2;
  }
  const int same = !memcmp(e, e_o, 2 * n) && !memcmp(qe, qe_o, 2 * n) && !memcmp(v, v_o, n) && !memcmp(q1, q1_o, 2 * n) &&
                   !memcmp(r1, r1_o, 2 * n) && !memcmp(q2, q2_o, n);
  printf("N=%d q=%d p=%d B=%lld kernel=%s: %s\n", N, q, p, (long long)B, ntru_engine_last_kernel(eng), same ? "identical" : "DIFFERENT");
  free(h); free(e); free(qe); free(q1); free(r1); free(e_o); free(qe_o); free(q1_o); free(r1_o);
  free(f); free(fp); free(r); free(m); free(v); free(q2); free(v_o); free(q2_o);
  return same ? 0 : 1;
}

int main(void) {
  ntru_engine_t *eng = NULL;
  CHECK(ntru_engine_create(0, &eng));
  int bad = 0;
  bad |= one_set(eng, 821, 4096, 3, 70);      /* BASELINE config 3: matrix-core kernels */
  bad |= one_set(eng, 701, 8192, 3, 33);
  bad |= one_set(eng, 167, 128, 3, 5);
  bad |= one_set(eng, 1279, 4096, 3, 9);      /* N > 1024: vector-ALU kernels */
  /* error path: a modulus the kernels do not implement must be refused, with a message */
  uint16_t dummy16[4] = {0}; uint8_t dummy8[4] = {0};
  if (ntru_encrypt_batch(eng, 4, 12, dummy16, dummy8, dummy8, 1, dummy16, NULL) == NTRU_OK || !ntru_last_error()[0]) {
    fprintf(stderr, "q = 12 was not refused\n");
    bad |= 1;
  }
  /* page-locked buffers from the engine: the same encrypt, every array DMA'd in place, must give the same bytes */
  {
    const int N = 509, q = 2048; const int64_t B = 40000;                 /* four chunks through the two slots */
    const size_t n = (size_t)B * N;
    uint16_t *h = malloc(2 * (size_t)N), *e1 = malloc(2 * n);
    uint8_t *r = ntru_host_alloc(n), *m = ntru_host_alloc(n);
    uint16_t *e2 = ntru_host_alloc(2 * n), *qe2 = ntru_host_alloc(2 * n);
    if (!r || !m || !e2 || !qe2) { fprintf(stderr, "ntru_host_alloc failed: %s\n", ntru_last_error()); return 2; }
    for (int i = 0; i < N; i++) h[i] = (uint16_t)(rnd() % (uint32_t)q);
    for (size_t i = 0; i < n; i++) { r[i] = (uint8_t)(rnd() % 3); m[i] = (uint8_t)(rnd() & 1); }
    CHECK(ntru_encrypt_batch(eng, N, q, h, r, m, B, e2, qe2));
    uint8_t *rp = malloc(n), *mp = malloc(n);
    memcpy(rp, r, n); memcpy(mp, m, n);
    CHECK(ntru_encrypt_batch(eng, N, q, h, rp, mp, B, e1, NULL));           /* pageable, value-only */
    const int same = !memcmp(e1, e2, 2 * n);
    printf("pinned vs pageable host buffers, N=%d B=%lld: %s\n", N, (long long)B, same ? "identical" : "DIFFERENT");
    bad |= !same;
    ntru_host_free(r); ntru_host_free(m); ntru_host_free(e2); ntru_host_free(qe2); free(h); free(e1); free(rp); free(mp);
  }
  /* the device-resident pipeline from C: a given r (so that the oracle can follow), encrypt -> decrypt on the GPU, only value
   * and the ciphertexts come back; then plain device buffers through the *_dev entry points */
  {
    const int N = 821, q = 4096, p = 3; const int64_t B = 9001;           /* four chunks */
    const size_t n = (size_t)B * N;
    uint16_t *h = malloc(2 * (size_t)N), *e = malloc(2 * n), *e_o = malloc(2 * n), *t16 = malloc(2 * n);
    int8_t *f = malloc(N);
    uint8_t *fp = malloc(N), *r = malloc(n), *m = malloc(n), *v = malloc(n), *v_o = malloc(n), *t8 = malloc(n);
    for (int i = 0; i < N; i++) { h[i] = (uint16_t)(rnd() % (uint32_t)q); f[i] = (int8_t)(rnd() % 3) - 1; fp[i] = (uint8_t)(rnd() % 3); }
    for (size_t i = 0; i < n; i++) { r[i] = (uint8_t)(rnd() % 3); m[i] = (uint8_t)(rnd() & 1); }
    CHECK(ntru_pipeline_batch(eng, N, q, p, h, f, fp, NULL, 0, 0, 0, r, m, B, NULL, e, v, NULL));
    if (orc_encrypt_batch(N, q, h, r, m, B, e_o, NULL, 0) || orc_decrypt_batch(N, q, p, f, fp, e_o, B, v_o, t16, t16, t8, 0)) return 2;
    int same = !memcmp(e, e_o, 2 * n) && !memcmp(v, v_o, n);
    void *d_h, *d_r, *d_m, *d_e;
    CHECK(ntru_dev_alloc(eng, 2 * (size_t)N, &d_h)); CHECK(ntru_dev_alloc(eng, n, &d_r)); CHECK(ntru_dev_alloc(eng, n, &d_m));
    CHECK(ntru_dev_alloc(eng, 2 * n, &d_e));
    CHECK(ntru_dev_upload(eng, d_h, h, 2 * (size_t)N)); CHECK(ntru_dev_upload(eng, d_r, r, n)); CHECK(ntru_dev_upload(eng, d_m, m, n));
    CHECK(ntru_encrypt_batch_dev(eng, N, q, d_h, d_r, d_m, B, d_e, NULL));
    memset(e, 0, 2 * n);
    CHECK(ntru_dev_download(eng, e, d_e, 2 * n));
    same = same && !memcmp(e, e_o, 2 * n);
    CHECK(ntru_dev_free(eng, d_h)); CHECK(ntru_dev_free(eng, d_r)); CHECK(ntru_dev_free(eng, d_m)); CHECK(ntru_dev_free(eng, d_e));
    printf("pipeline + device buffers, N=%d B=%lld: %s\n", N, (long long)B, same ? "identical" : "DIFFERENT");
    bad |= !same;
    free(h); free(e); free(e_o); free(t16); free(f); free(fp); free(r); free(m); free(v); free(v_o); free(t8);
  }
  /* generic family: the two divisions of test/circuits.test.js:165-170 that the reference computes / refuses, and its
   * worked Euclid example (index.js:411-423): [4,2,0,3]^-1 mod [3,2,1] over Z_11 = [5,8] */
  {
    const int64_t a1[3] = {81, 2, 96}, b1[3] = {48, 2, 31}, a2[2] = {1, 2}, b2[2] = {2, 3}, a3[4] = {4, 2, 0, 3}, b3[3] = {3, 2, 1};
    int64_t quot[32], rem[32], gcd[32], inv[32]; int32_t ql, rl, gl, il; uint8_t st;
    CHECK(ntru_generic_divide(eng, 3, 3, 128, a1, b1, 1, quot, &ql, rem, &rl, &st));
    const int ok1 = st == 0 && ql == 1 && quot[0] == 32 && rl == 2 && rem[0] == 81 && rem[1] == 66;
    CHECK(ntru_generic_divide(eng, 2, 2, 3, a2, b2, 1, quot, &ql, rem, &rl, &st));
    const int ok2 = st == NTRU_GENERIC_NO_INVERSE;
    CHECK(ntru_generic_eea(eng, 4, 3, 11, a3, b3, 1, gcd, &gl, inv, &il, &st));
    const int ok3 = st == 0 && gl == 1 && gcd[0] == 1 && il == 2 && inv[0] == 5 && inv[1] == 8;
    printf("generic family: divide %s, refused divide %s, Euclid example %s\n", ok1 ? "ok" : "WRONG", ok2 ? "ok" : "WRONG", ok3 ? "ok" : "WRONG");
    bad |= !(ok1 && ok2 && ok3);
  }
  ntru_engine_destroy(eng);
  return bad;
}
