/*
 * ntru_oracle.c -- CPU restatement of the numtel/ntru-circom hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The shipped path
 * (ntru-circom_amd/) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function here is checked by tests/test_oracle_golden.py
 * against vectors captured from the unmodified reference running under Node in the
 * build container (tests/golden/gen_golden.mjs -> tests/golden/ JSON files).
 *
 * Two flavours of the arithmetic are provided:
 *   mode ORC_EXACT (0)    exact-integer schoolbook product + closed-form split by 1-x^N
 *                         (identical results, far less work)
 *   mode ORC_FAITHFUL (1) the reference's own algorithm: double-precision complex
 *                         radix-2 FFT product (index.js:277-355), generic long
 *                         division (index.js:358-401) whose every step re-derives
 *                         the inverse of the divisor's lead by brute force
 *                         (index.js:224-232).  This is the "reference-equivalent"
 *                         CPU baseline that bench.py times.
 *
 * Coefficients are int64_t in the generic (variable-length) functions because the
 * reference works on JS Numbers that may be negative or unreduced; the fixed-stride
 * batch entry points at the bottom use the same flat uint16/uint8 layout as the
 * engine's C ABI (include/ntru_engine.h) so outputs can be compared byte for byte.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_DIV_ZERO 1    /* "Cannot divide by zero polynomial."  index.js:360 */
#define ORC_ERR_NO_INVERSE 2  /* "No inverse exists for division."    index.js:378 */
#define ORC_ERR_SAMPLER 3     /* "The total of 1s and -1s cannot exceed the array length." index.js:463 */
#define ORC_ERR_ARG 4

#define ORC_EXACT 0
#define ORC_FAITHFUL 1

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* JS `((x % p) + p) % p` with C's truncating %, same as JS's. */
static inline int64_t posmod(int64_t x, int64_t p) { return ((x % p) + p) % p; }

/* index.js:210-215 -- highest index holding a non-zero value, -1 if none. */
int orc_degree(const int64_t *a, int n) {
  for (int i = n - 1; i >= 0; i--)
    if (a[i] != 0) return i;
  return -1;
}

/* index.js:218-221 -- keep [0..degree]; the zero polynomial becomes [0].  In place; returns new length. */
int orc_trim(int64_t *a, int n) {
  int d = orc_degree(a, n);
  if (d < 0) { a[0] = 0; return 1; }
  return d + 1;
}

/* index.js:224-232 -- brute-force search; -1 stands for the reference's `null`. */
int64_t orc_mod_inverse(int64_t a, int64_t p) {
  a = posmod(a, p);
  for (int64_t x = 1; x < p; x++)
    if ((a * x) % p == 1) return x;
  return -1;
}

/* index.js:235-244 -- coefficient-wise (a+b) into [0,p), trimmed.  out needs max(la,lb,1) slots. */
int orc_add(const int64_t *a, int la, const int64_t *b, int lb, int64_t p, int64_t *out) {
  int n = la > lb ? la : lb;
  for (int i = 0; i < n; i++) {
    int64_t x = i < la ? a[i] : 0, y = i < lb ? b[i] : 0;
    out[i] = posmod(x + y, p);
  }
  if (n == 0) { out[0] = 0; return 1; }
  return orc_trim(out, n);
}

/* ---- reference-equivalent product: complex FFT in doubles (index.js:277-355) ------------- */

static void fft_inplace(double *re, double *im, int n, int invert) {
  /* bit-reversal permutation (index.js:280-292) */
  int bits = 0;
  while ((1 << bits) < n) bits++;
  for (int i = 0; i < n; i++) {
    int j = 0;
    for (int b = 0; b < bits; b++)
      if (i & (1 << b)) j |= 1 << (bits - 1 - b);
    if (i < j) {
      double t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  /* butterflies, twiddle advanced by repeated multiplication exactly as index.js:295-308 does,
     so the rounding sequence is the reference's */
  for (int len = 2; len <= n; len <<= 1) {
    double ang = (2 * M_PI / len) * (invert ? -1 : 1);
    double wlr = cos(ang), wli = sin(ang);
    int half = len / 2;
    for (int base = 0; base < n; base += len) {
      double wr = 1, wi = 0;
      for (int j = 0; j < half; j++) {
        int lo = base + j, hi = lo + half;
        double vr = re[hi] * wr - im[hi] * wi;
        double vi = re[hi] * wi + im[hi] * wr;
        double ur = re[lo], ui = im[lo];
        re[lo] = ur + vr; im[lo] = ui + vi;
        re[hi] = ur - vr; im[hi] = ui - vi;
        double nwr = wr * wlr - wi * wli;
        double nwi = wr * wli + wi * wlr;
        wr = nwr; wi = nwi;
      }
    }
  }
  if (invert)
    for (int i = 0; i < n; i++) { re[i] /= n; im[i] /= n; }
}

/* index.js:319-355.  out needs max(la+lb-1,1) slots; returns trimmed length. */
int orc_multiply_fft(const int64_t *a, int la, const int64_t *b, int lb, int64_t p, int64_t *out) {
  if (la == 0 || lb == 0) { out[0] = 0; return 1; } /* index.js:320 */
  int rl = la + lb - 1, n = 1;
  while (n < rl) n <<= 1;
  double *buf = (double *)calloc((size_t)4 * n, sizeof(double));
  double *ar = buf, *ai = buf + n, *br = buf + 2 * n, *bi = buf + 3 * n;
  for (int i = 0; i < la; i++) ar[i] = (double)a[i];
  for (int i = 0; i < lb; i++) br[i] = (double)b[i];
  fft_inplace(ar, ai, n, 0);
  fft_inplace(br, bi, n, 0);
  for (int i = 0; i < n; i++) {
    double r = ar[i] * br[i] - ai[i] * bi[i];
    double m = ar[i] * bi[i] + ai[i] * br[i];
    ar[i] = r; ai[i] = m;
  }
  fft_inplace(ar, ai, n, 1);
  for (int i = 0; i < rl; i++) {
    /* JS Math.round = floor(x + 0.5) */
    int64_t v = (int64_t)floor(ar[i] + 0.5);
    out[i] = posmod(v, p);
  }
  free(buf);
  return orc_trim(out, rl);
}

/* Exact-integer restatement of the same function (equal on every config). */
int orc_multiply_exact(const int64_t *a, int la, const int64_t *b, int lb, int64_t p, int64_t *out) {
  if (la == 0 || lb == 0) { out[0] = 0; return 1; }
  int rl = la + lb - 1;
  for (int k = 0; k < rl; k++) out[k] = 0;
  for (int i = 0; i < la; i++) {
    int64_t x = a[i];
    if (x == 0) continue;
    for (int j = 0; j < lb; j++) out[i + j] += x * b[j];
  }
  for (int k = 0; k < rl; k++) out[k] = posmod(out[k], p);
  return orc_trim(out, rl);
}

/* ---- reference-equivalent division: generic long division (index.js:358-401) --------------
 * quot needs max(la,1) slots, rem needs max(la, lb, 1)+1 slots.  Lengths come back trimmed. */
int orc_divide_long(const int64_t *a, int la, const int64_t *b, int lb, int64_t p,
                    int64_t *quot, int *lq, int64_t *rem, int *lr) {
  int degb = orc_degree(b, lb);
  if (degb < 0) return ORC_ERR_DIV_ZERO;
  int cap = (la > lb ? la : lb) + 1;
  for (int i = 0; i < cap; i++) rem[i] = i < la ? a[i] : 0;
  int nrem = la;                                   /* JS array length of `dividend` */
  int dega = orc_degree(a, la);
  int nq = dega - degb + 1;
  if (nq < 0) nq = 0;
  for (int i = 0; i < nq; i++) quot[i] = 0;
  for (;;) {
    int d = orc_degree(rem, nrem);                 /* rescanned every step, as index.js:372-373 */
    if (d < degb) break;
    int64_t lead = rem[d];
    int64_t inv = orc_mod_inverse(b[degb], p);     /* recomputed every step, index.js:376 */
    if (inv < 0) return ORC_ERR_NO_INVERSE;
    int64_t coeff = (lead * inv) % p;
    int shift = d - degb;
    quot[shift] = coeff;
    for (int i = 0; i <= degb; i++) {
      int idx = i + shift;
      int64_t v = (rem[idx] - coeff * b[i]) % p;
      if (v < 0) v += p;
      rem[idx] = v;
      if (idx + 1 > nrem) nrem = idx + 1;
    }
  }
  if (nq == 0) { quot[0] = 0; *lq = 1; } else *lq = orc_trim(quot, nq);
  if (nrem == 0) { rem[0] = 0; *lr = 1; } else *lr = orc_trim(rem, nrem);
  return ORC_OK;
}

/* Closed form of the same division when b = I = 1 - x^N and every a[i] is already in [0,p)
 * (SURVEY.md 0.3).  Valid for la <= 2N.  quot needs N slots, rem needs max(la,N,1) slots. */
int orc_divide_by_I(const int64_t *a, int la, int N, int64_t p,
                    int64_t *quot, int *lq, int64_t *rem, int *lr) {
  if (la > 2 * N) return ORC_ERR_ARG;
  int nlow = la < N ? la : N;
  for (int k = 0; k < nlow; k++) {
    int64_t hi = (N + k) < la ? a[N + k] : 0;
    rem[k] = (a[k] + hi) % p;
  }
  if (nlow == 0) { rem[0] = 0; *lr = 1; } else *lr = orc_trim(rem, nlow);
  int nq = la - N;
  if (nq <= 0) { quot[0] = 0; *lq = 1; return ORC_OK; }
  for (int k = 0; k < nq; k++) quot[k] = (p - a[N + k]) % p;
  *lq = orc_trim(quot, nq);
  return ORC_OK;
}

/* index.js:461-488 -- ternary sampler replayed from an explicit tape of u32 draws
 * (exactly len-1 draws are consumed, i descending, j = u32 % (i+1)). */
int orc_generate_custom_array(int len, int n1, int nm1, const uint32_t *draws, int64_t *out) {
  if (n1 + nm1 > len) return ORC_ERR_SAMPLER;
  for (int i = 0; i < len; i++) out[i] = 0;
  for (int i = 0; i < n1; i++) out[i] = 1;
  for (int i = n1; i < n1 + nm1; i++) out[i] = -1;
  int t = 0;
  for (int i = len - 1; i > 0; i--) {
    int j = (int)(draws[t++] % (uint32_t)(i + 1));
    int64_t tmp = out[i]; out[i] = out[j]; out[j] = tmp;
  }
  return ORC_OK;
}

/* index.js:201-206 */
int orc_calc_nbits(int64_t mod, int N) { return (int)ceil(log2((double)mod * (double)mod * (double)N)); }

/* ---- one polymul + split in either flavour ----------------------------------------------- */

/* Variable-length single-item entry used by the Python oracle wrapper: prod = a*b mod `mod`,
 * optionally + addend (the `m + r*h` of index.js:91), then split by I. */
int orc_polymul_split(const int64_t *a, int la, const int64_t *b, int lb,
                      const int64_t *addend, int ladd, int N, int64_t mod, int mode,
                      int64_t *quot, int *lq, int64_t *rem, int *lr) {
  int cap = 2 * N + 4 + la + lb + ladd;
  int64_t *prod = (int64_t *)calloc((size_t)3 * cap + N + 1, sizeof(int64_t));
  int64_t *sum = prod + cap, *remw = sum + cap, *Ipoly = remw + cap;
  Ipoly[0] = 1; Ipoly[N] = -1;                            /* index.js:25-27 */
  int lp = mode == ORC_FAITHFUL ? orc_multiply_fft(a, la, b, lb, mod, prod)
                                : orc_multiply_exact(a, la, b, lb, mod, prod);
  const int64_t *dvd = prod;
  if (addend) { lp = orc_add(addend, ladd, prod, lp, mod, sum); dvd = sum; }
  int rc = mode == ORC_FAITHFUL ? orc_divide_long(dvd, lp, Ipoly, N + 1, mod, quot, lq, remw, lr)
                                : orc_divide_by_I(dvd, lp, N, mod, quot, lq, remw, lr);
  if (rc == ORC_OK) memcpy(rem, remw, sizeof(int64_t) * (size_t)*lr);
  free(prod);
  return rc;
}

/* ---- fixed-stride batch entry points (same flat layout as include/ntru_engine.h) ---------- */

typedef struct {
  int N;
  int64_t *a, *b, *add, *prod, *sum, *quot, *rem, *Ipoly;
} work_t;

static void work_init(work_t *w, int N) {
  int cap = 2 * N + 8;
  w->N = N;
  w->a = (int64_t *)calloc((size_t)8 * cap, sizeof(int64_t));
  w->b = w->a + cap; w->add = w->b + cap; w->prod = w->add + cap; w->sum = w->prod + cap;
  w->quot = w->sum + cap; w->rem = w->quot + cap; w->Ipoly = w->rem + cap;
  w->Ipoly[0] = 1; w->Ipoly[N] = -1;
}
static void work_free(work_t *w) { free(w->a); }

static int trimmed_len_i64(const int64_t *a, int n) {
  int d = orc_degree(a, n);
  return d < 0 ? 1 : d + 1;
}

/* prod = a*b (+add) then split; results zero-padded to N into q_out/r_out. `la`/`lb` are the lengths
 * the reference would see (trimmed keys, untrimmed r). */
static int item_mul_split(work_t *w, int la, int lb, int ladd, int64_t mod, int mode,
                          int64_t *q_out, int64_t *r_out, int *lq_out, int *lr_out) {
  int N = w->N, lq, lr, lp;
  lp = mode == ORC_FAITHFUL ? orc_multiply_fft(w->a, la, w->b, lb, mod, w->prod)
                            : orc_multiply_exact(w->a, la, w->b, lb, mod, w->prod);
  const int64_t *dvd = w->prod;
  if (ladd >= 0) { lp = orc_add(w->add, ladd, w->prod, lp, mod, w->sum); dvd = w->sum; }
  int rc = mode == ORC_FAITHFUL ? orc_divide_long(dvd, lp, w->Ipoly, N + 1, mod, w->quot, &lq, w->rem, &lr)
                                : orc_divide_by_I(dvd, lp, N, mod, w->quot, &lq, w->rem, &lr);
  if (rc) return rc;
  for (int k = 0; k < N; k++) {
    q_out[k] = k < lq ? w->quot[k] : 0;
    r_out[k] = k < lr ? w->rem[k] : 0;
  }
  if (lq_out) *lq_out = lq;
  if (lr_out) *lr_out = lr;
  return ORC_OK;
}

/* generic a*b mod `mod` split by 1-x^N, per-item operands (the engine's ntru_polymul_split). */
int orc_polymul_split_batch(int N, int mod, const uint16_t *a, const uint16_t *b, int64_t B,
                            uint16_t *quot, uint16_t *rem, int mode) {
  work_t w; work_init(&w, N);
  int64_t *qo = (int64_t *)calloc((size_t)2 * N, sizeof(int64_t)), *ro = qo + N;
  int rc = ORC_OK;
  for (int64_t it = 0; it < B && !rc; it++) {
    for (int k = 0; k < N; k++) { w.a[k] = a[it * N + k]; w.b[k] = b[it * N + k]; }
    rc = item_mul_split(&w, trimmed_len_i64(w.a, N), trimmed_len_i64(w.b, N), -1, mod, mode, qo, ro, 0, 0);
    for (int k = 0; k < N; k++) { quot[it * N + k] = (uint16_t)qo[k]; rem[it * N + k] = (uint16_t)ro[k]; }
  }
  free(qo); work_free(&w);
  return rc;
}

/* index.js:87-110.  r in {0,1,2} (the -1 -> p-1 map of :89 already applied), m any bytes (README: 0/1/2),
 * h in [0,q).  e = remainderE, quotE = quotientE (may be NULL); both stride N (their [N]-th entry is always 0). */
int orc_encrypt_batch(int N, int q, const uint16_t *h, const uint8_t *r, const uint8_t *m, int64_t B,
                      uint16_t *e, uint16_t *quotE, int mode) {
  work_t w; work_init(&w, N);
  int64_t *qo = (int64_t *)calloc((size_t)2 * N, sizeof(int64_t)), *ro = qo + N;
  int rc = ORC_OK;
  for (int64_t it = 0; it < B && !rc; it++) {
    for (int k = 0; k < N; k++) { w.a[k] = r[it * N + k]; w.b[k] = h[k]; w.add[k] = m[it * N + k]; }
    /* r is never trimmed by the reference (index.js:89-90); h is stored trimmed (index.js:78) */
    rc = item_mul_split(&w, N, trimmed_len_i64(w.b, N), N, q, mode, qo, ro, 0, 0);
    for (int k = 0; k < N; k++) {
      e[it * N + k] = (uint16_t)ro[k];
      if (quotE) quotE[it * N + k] = (uint16_t)qo[k];
    }
  }
  free(qo); work_free(&w);
  return rc;
}

/* index.js:111-140.  f in {-1,0,1}, fp in {0,1,2}, e in [0,q).  value = remainder2 (stride N).
 * quot1/rem1/quot2 may be NULL. */
int orc_decrypt_batch(int N, int q, int p, const int8_t *f, const uint8_t *fp, const uint16_t *e, int64_t B,
                      uint8_t *value, uint16_t *quot1, uint16_t *rem1, uint8_t *quot2, int mode) {
  work_t w; work_init(&w, N);
  int64_t *qo = (int64_t *)calloc((size_t)4 * N, sizeof(int64_t)), *ro = qo + N, *q2 = ro + N, *r2 = q2 + N;
  int rc = ORC_OK;
  int lfp;
  { for (int k = 0; k < N; k++) w.a[k] = fp[k]; lfp = trimmed_len_i64(w.a, N); }
  for (int64_t it = 0; it < B && !rc; it++) {
    int lr1;
    for (int k = 0; k < N; k++) { w.a[k] = f[k] == -1 ? q - 1 : f[k]; w.b[k] = e[it * N + k]; }  /* :112 */
    rc = item_mul_split(&w, N, trimmed_len_i64(w.b, N), -1, q, mode, qo, ro, 0, &lr1);          /* :113-114 */
    if (rc) break;
    /* centred lift on the TRIMMED remainder, strict >, index.js:117 (kept verbatim incl. its q=1 mod 3 quirk) */
    for (int k = 0; k < lr1; k++) { int64_t x = ro[k]; w.b[k] = 2 * x > q ? (x + 1) % p : x % p; }
    for (int k = 0; k < N; k++) w.a[k] = fp[k];
    rc = item_mul_split(&w, lfp, lr1, -1, p, mode, q2, r2, 0, 0);                                /* :118-119 */
    for (int k = 0; k < N; k++) {
      value[it * N + k] = (uint8_t)r2[k];
      if (quot1) quot1[it * N + k] = (uint16_t)qo[k];
      if (rem1) rem1[it * N + k] = (uint16_t)ro[k];
      if (quot2) quot2[it * N + k] = (uint8_t)q2[k];
    }
  }
  free(qo); work_free(&w);
  return rc;
}

/* index.js:141-197, per-item key material.  flags bit0/1/2 = the reference would throw
 * 'invalid fq' / 'invalid fp' / 'invalid h' (checks of :159,:162,:165 reproduced literally). */
int orc_verify_keys_batch(int N, int q, int p, const int8_t *f, const int8_t *g, const uint16_t *fq,
                          const uint8_t *fp, const uint16_t *h, int64_t B,
                          uint16_t *quot_fq, uint16_t *rem_fq, uint8_t *quot_fp, uint8_t *rem_fp,
                          uint16_t *quot_h, uint16_t *rem_h, uint8_t *flags, int mode) {
  work_t w; work_init(&w, N);
  int64_t *qo = (int64_t *)calloc((size_t)2 * N, sizeof(int64_t)), *ro = qo + N;
  int rc = ORC_OK;
  for (int64_t it = 0; it < B && !rc; it++) {
    const int8_t *fi = f + it * N, *gi = g + it * N;
    const uint16_t *fqi = fq + it * N, *hi = h + it * N;
    const uint8_t *fpi = fp + it * N;
    int lr, fl = 0, lfq, lfp;
    /* fq * fmodq mod q  (:158) -- f itself is never trimmed (length N array) */
    for (int k = 0; k < N; k++) { w.a[k] = fqi[k]; w.b[k] = fi[k] == -1 ? q - 1 : fi[k]; }
    lfq = trimmed_len_i64(w.a, N);
    rc = item_mul_split(&w, lfq, N, -1, q, mode, qo, ro, 0, &lr); if (rc) break;
    if (lr != 1 && ro[0] != 1) fl |= 1;
    for (int k = 0; k < N; k++) { quot_fq[it * N + k] = (uint16_t)qo[k]; rem_fq[it * N + k] = (uint16_t)ro[k]; }
    /* fp * fmodp mod p  (:161) */
    for (int k = 0; k < N; k++) { w.a[k] = fpi[k]; w.b[k] = fi[k] == -1 ? p - 1 : fi[k]; }
    lfp = trimmed_len_i64(w.a, N);
    rc = item_mul_split(&w, lfp, N, -1, p, mode, qo, ro, 0, &lr); if (rc) break;
    if (lr != 1 && ro[0] != 1) fl |= 2;
    for (int k = 0; k < N; k++) { quot_fp[it * N + k] = (uint8_t)qo[k]; rem_fp[it * N + k] = (uint8_t)ro[k]; }
    /* (p*fq, unreduced) * g mod q  (:155,:164) */
    for (int k = 0; k < N; k++) { w.a[k] = (int64_t)fqi[k] * p; w.b[k] = gi[k] == -1 ? q - 1 : gi[k]; }
    rc = item_mul_split(&w, lfq, N, -1, q, mode, qo, ro, 0, &lr); if (rc) break;
    {
      int lh = 1;
      for (int k = N - 1; k >= 0; k--) if (hi[k]) { lh = k + 1; break; }
      for (int k = 0; k < lh; k++)
        if (k >= lr || ro[k] != (int64_t)hi[k]) { fl |= 4; break; }           /* :165 */
    }
    for (int k = 0; k < N; k++) { quot_h[it * N + k] = (uint16_t)qo[k]; rem_h[it * N + k] = (uint16_t)ro[k]; }
    flags[it] = (uint8_t)fl;
  }
  free(qo); work_free(&w);
  return rc;
}

/* ---- replayable draw stream for the on-device sampler ---------------------------------------
 * The reference draws one u32 per Fisher-Yates step from crypto.getRandomValues (index.js:481-483).  The engine's
 * sampler replaces that CSPRNG by the ChaCha20 keystream (RFC 8439 block function) of a caller-supplied 256-bit key:
 * item b uses nonce = (b_lo, b_hi, 0x4e545255 "NTRU") and block counter 0,1,2,...; draw t of item b is keystream word
 * t (little-endian).  Everything else is generateCustomArray verbatim (orc_generate_custom_array above). */
static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);

/* rounds: 20 (RFC 8439), 12 or 8 -- the engine's ntru_engine_set_sampler_rounds; rounds / 2 double rounds */
void orc_chacha_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], int rounds, uint32_t out[16]) {
  uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                    key[4], key[5], key[6], key[7], counter, nonce[0], nonce[1], nonce[2]};
  uint32_t x[16];
  for (int i = 0; i < 16; i++) x[i] = s[i];
  for (int r = 0; r < rounds / 2; r++) {
    QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
    QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
void orc_chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint32_t out[16]) {
  orc_chacha_block(key, counter, nonce, 20, out);
}

/* draws[t], t = 0..n-1, of item `item` */
void orc_draw_stream_rounds(const uint32_t key[8], uint64_t item, int n, int rounds, uint32_t *draws) {
  uint32_t nonce[3] = {(uint32_t)item, (uint32_t)(item >> 32), 0x4e545255u}, blk[16];
  for (int t = 0; t < n; t++) {
    if ((t & 15) == 0) orc_chacha_block(key, (uint32_t)(t >> 4), nonce, rounds, blk);
    draws[t] = blk[t & 15];
  }
}
void orc_draw_stream(const uint32_t key[8], uint64_t item, int n, uint32_t *draws) { orc_draw_stream_rounds(key, item, n, 20, draws); }

/* r rows for items first_item .. first_item+B-1: n1 ones, n2 entries equal to `other`, zeros; stride N. */
int orc_sample_ternary_batch_rounds(int N, int n1, int n2, int other, const uint32_t key[8], uint64_t first_item, int64_t B,
                                    int rounds, uint8_t *out) {
  if (n1 + n2 > N) return ORC_ERR_SAMPLER;
  if (rounds != 20 && rounds != 12 && rounds != 8) return ORC_ERR_ARG;
  uint32_t *draws = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(N > 1 ? N - 1 : 1));
  int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
  for (int64_t b = 0; b < B; b++) {
    orc_draw_stream_rounds(key, first_item + (uint64_t)b, N - 1, rounds, draws);
    orc_generate_custom_array(N, n1, n2, draws, tmp);
    for (int k = 0; k < N; k++) out[b * N + k] = (uint8_t)(tmp[k] == -1 ? other : tmp[k]);
  }
  free(draws); free(tmp);
  return ORC_OK;
}
int orc_sample_ternary_batch(int N, int n1, int n2, int other, const uint32_t key[8], uint64_t first_item, int64_t B,
                             uint8_t *out) {
  return orc_sample_ternary_batch_rounds(N, n1, n2, other, key, first_item, B, 20, out);
}

/* ---- BN254 field-element packing, index.js:572-620 -------------------------------------------
 * A packed field element is stored as four little-endian uint64 limbs (values stay below 2^252). */
int orc_pack_params(int max_val, int data_len, int *bits, int *per, int *arr_len, int *out_size) {
  if (max_val < 1 || max_val > 65535 || data_len < 0) return ORC_ERR_ARG;
  int b = 0;
  while ((max_val >> b) != 0) b++;                   /* floor(log2(maxVal) + 1), index.js:573 */
  int n = 252 / b;                                   /* :574 */
  int al = ((data_len + n - 1) / n) * n;             /* :575-578 */
  if (al < 3 * n) al = 3 * n;
  int os = (al + n - 1) / n;                         /* :580 */
  if (os < 3) os = 3;
  *bits = b; *per = n; *arr_len = al; *out_size = os;
  return ORC_OK;
}

static void limb_or(uint64_t *limbs, int bitpos, uint64_t v) {
  int l = bitpos >> 6, s = bitpos & 63;
  limbs[l] |= v << s;
  if (s && l + 1 < 4) limbs[l + 1] |= v >> (64 - s);
}

/* index.js:581-587.  data: [B][data_len]; out: [B][out_size][4]. */
int orc_pack_batch(int max_val, int data_len, const uint16_t *data, int64_t B, uint64_t *out) {
  int bits, per, al, os;
  int rc = orc_pack_params(max_val, data_len, &bits, &per, &al, &os);
  if (rc) return rc;
  memset(out, 0, sizeof(uint64_t) * 4 * (size_t)os * (size_t)B);
  for (int64_t b = 0; b < B; b++)
    for (int i = 0; i < data_len; i++)
      limb_or(out + ((size_t)b * os + i / per) * 4, (i % per) * bits, data[b * data_len + i]);
  return ORC_OK;
}

/* index.js:598-620 before trimming.  in: [B][packed_size][4]; out: [B][per * packed_size]. */
int orc_unpack_batch(int max_val, int packed_bits, const uint64_t *in, int packed_size, int64_t B, uint16_t *out) {
  if (max_val < 1 || max_val > 65535) return ORC_ERR_ARG;
  int bits = 0;
  while ((max_val >> bits) != 0) bits++;
  int per = packed_bits / bits;
  if (per < 1 || per * bits > 256) return ORC_ERR_ARG;
  for (int64_t b = 0; b < B; b++)
    for (int i = 0; i < packed_size; i++) {
      const uint64_t *l = in + ((size_t)b * packed_size + i) * 4;
      for (int j = 0; j < per; j++) {
        int pos = j * bits, w = pos >> 6, s = pos & 63;
        uint64_t v = l[w] >> s;
        if (s + bits > 64 && w + 1 < 4) v |= l[w + 1] << (64 - s);
        out[((size_t)b * packed_size + i) * per + j] = (uint16_t)(v & ((1u << bits) - 1));
      }
    }
  return ORC_OK;
}
