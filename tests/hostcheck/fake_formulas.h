// The test-double "kernels" (shared by fake_device.cpp, which runs them on a stream worker, and by the driver, which runs them on the
// whole batch at once to get the expected values): every output element depends on every input array, the key rows and its position.
#ifndef NTRU_FAKE_FORMULAS_H
#define NTRU_FAKE_FORMULAS_H
#include <cstdint>

static inline void fake_encrypt(int N, int q, const uint16_t *h, const uint8_t *r, const uint8_t *m, int64_t B, uint16_t *e, uint16_t *quot) {
  for (int64_t b = 0; b < B; b++)
    for (int i = 0; i < N; i++) {
      const uint32_t v = (uint32_t)r[b * N + i] * 5u + m[b * N + i] * 3u + h[i] + (uint32_t)i;
      e[b * N + i] = (uint16_t)(v & (uint32_t)(q - 1));
      if (quot) quot[b * N + i] = (uint16_t)((v * 7u + 1u) & (uint32_t)(q - 1));
    }
}
static inline void fake_decrypt(int N, int q, int p, const int8_t *f, const uint8_t *fp, const uint16_t *e, int64_t B, uint8_t *value,
                                uint16_t *quot1, uint16_t *rem1, uint8_t *quot2) {
  for (int64_t b = 0; b < B; b++)
    for (int i = 0; i < N; i++) {
      const uint32_t v = (uint32_t)e[b * N + i] + (uint32_t)(f[i] + 1) * 11u + fp[i] * 13u + (uint32_t)i;
      value[b * N + i] = (uint8_t)(v % (uint32_t)p);
      if (quot1) quot1[b * N + i] = (uint16_t)((v + 1u) & (uint32_t)(q - 1));
      if (rem1) rem1[b * N + i] = (uint16_t)((v + 2u) & (uint32_t)(q - 1));
      if (quot2) quot2[b * N + i] = (uint8_t)((v + 3u) % (uint32_t)p);
    }
}
static inline void fake_polymul(int N, int mod, const uint16_t *a, const uint16_t *b2, int64_t B, uint16_t *quot, uint16_t *rem) {
  for (int64_t i = 0; i < B * N; i++) { quot[i] = (uint16_t)(((uint32_t)a[i] * 3u + b2[i]) % (uint32_t)mod); rem[i] = (uint16_t)(((uint32_t)a[i] + b2[i] * 5u + 1u) % (uint32_t)mod); }
}
static inline void fake_public_key(int N, int q, int p, const uint16_t *fq, const int8_t *g, int64_t B, uint16_t *h) {
  for (int64_t i = 0; i < B * N; i++) h[i] = (uint16_t)(((uint32_t)fq[i] * (uint32_t)p + (uint32_t)(g[i] + 1) * 9u) & (uint32_t)(q - 1));
}
static inline void fake_verify(int N, int q, int p, const int8_t *f, const int8_t *g, const uint16_t *fq, const uint8_t *fp, const uint16_t *h,
                               int64_t B, uint16_t *qfq, uint16_t *rfq, uint8_t *qfp, uint8_t *rfp, uint16_t *qh, uint16_t *rh, uint8_t *flags) {
  for (int64_t b = 0; b < B; b++) {
    for (int i = 0; i < N; i++) {
      const int64_t k = b * N + i;
      const uint32_t v = (uint32_t)(f[k] + 1) + (uint32_t)(g[k] + 1) * 3u + fq[k] + fp[k] * 7u + h[k];
      qfq[k] = (uint16_t)(v & (uint32_t)(q - 1)); rfq[k] = (uint16_t)((v + 1u) & (uint32_t)(q - 1));
      qfp[k] = (uint8_t)(v % (uint32_t)p); rfp[k] = (uint8_t)((v + 1u) % (uint32_t)p);
      qh[k] = (uint16_t)((v + 2u) & (uint32_t)(q - 1)); rh[k] = (uint16_t)((v + 3u) & (uint32_t)(q - 1));
    }
    flags[b] = (uint8_t)((fq[b * N] + h[b * N]) & 7);
  }
}
#endif
