/*
 * addon.c -- N-API binding of include/ntru_engine.h for Node.js (N-API v3+, works on Node 12).
 *
 * Thin by design: every exported function unpacks TypedArray arguments into the plain pointers the C ABI takes
 * (napi_get_typedarray_info), calls the engine, and throws a JS Error carrying ntru_last_error() on failure.
 * All reference-shaped behaviour (padding, trimming, {value, inputs, params} objects) lives in index.mjs.
 */
#define NAPI_VERSION 6
#include <node_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ntru_engine.h"

#define NAPI_OK(call)                                                              \
  do {                                                                             \
    if ((call) != napi_ok) {                                                       \
      napi_throw_error(env, NULL, "N-API call failed: " #call);                    \
      return NULL;                                                                 \
    }                                                                              \
  } while (0)

static ntru_engine_t *g_engine = NULL;
static ntru_multi_t *g_multi = NULL;     /* useDevices([...]): the batch entry points shard over these devices instead */
/* One engine, one caller at a time: the *Async entry points run on libuv worker threads, everything else on the JS thread. */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
#define ENGINE_CALL(rc, expr) do { pthread_mutex_lock(&g_lock); (rc) = (expr); pthread_mutex_unlock(&g_lock); } while (0)

static napi_value throw_engine(napi_env env, int rc) {
  char buf[512];
  snprintf(buf, sizeof buf, "ntru engine error %d: %s", rc, ntru_last_error());
  napi_throw_error(env, NULL, buf);
  return NULL;
}

static int get_i32(napi_env env, napi_value v, int32_t *out) { return napi_get_value_int32(env, v, out) == napi_ok; }

/* Returns the data pointer of a TypedArray of the wanted element type holding at least `need` elements;
 * null/undefined gives NULL when `optional`. */
static int get_buf(napi_env env, napi_value v, napi_typedarray_type want, size_t need, int optional, void **out) {
  napi_valuetype vt;
  *out = NULL;
  if (napi_typeof(env, v, &vt) != napi_ok) return 0;
  if (vt == napi_null || vt == napi_undefined) return optional;
  bool is_ta = false;
  if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) return 0;
  napi_typedarray_type t; size_t len; void *data;
  if (napi_get_typedarray_info(env, v, &t, &len, &data, NULL, NULL) != napi_ok) return 0;
  if (t != want || len < need) return 0;
  *out = data;
  return 1;
}

#define ARGS(n)                                                                    \
  size_t argc = (n);                                                               \
  napi_value argv[(n)];                                                            \
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));                   \
  if (argc < (n)) { napi_throw_type_error(env, NULL, "too few arguments"); return NULL; }

#define BAD_ARGS() do { napi_throw_type_error(env, NULL, "bad argument types / sizes"); return NULL; } while (0)

static napi_value undefined(napi_env env) { napi_value u; napi_get_undefined(env, &u); return u; }

static int ensure_engine(napi_env env) {
  if (g_engine) return 1;
  napi_throw_error(env, NULL, "ntru engine not created: call create(device) first (there is no CPU fallback)");
  return 0;
}

static napi_value DeviceCount(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value r; NAPI_OK(napi_create_int32(env, ntru_engine_device_count(), &r)); return r;
}

static napi_value Create(napi_env env, napi_callback_info info) {
  ARGS(1)
  int32_t dev;
  if (!get_i32(env, argv[0], &dev)) BAD_ARGS();
  pthread_mutex_lock(&g_lock);
  if (g_engine) { ntru_engine_destroy(g_engine); g_engine = NULL; }
  int rc = ntru_engine_create(dev, &g_engine);
  pthread_mutex_unlock(&g_lock);
  if (rc) return throw_engine(env, rc);
  return undefined(env);
}

static napi_value Destroy(napi_env env, napi_callback_info info) {
  (void)info;
  pthread_mutex_lock(&g_lock);
  if (g_engine) { ntru_engine_destroy(g_engine); g_engine = NULL; }
  if (g_multi) { ntru_multi_destroy(g_multi); g_multi = NULL; }
  pthread_mutex_unlock(&g_lock);
  return undefined(env);
}

/* useDevices(ids:Int32Array) -> number of engines.  An empty array goes back to the single engine of create(). */
static napi_value UseDevices(napi_env env, napi_callback_info info) {
  ARGS(1)
  bool is_ta = false;
  if (napi_is_typedarray(env, argv[0], &is_ta) != napi_ok || !is_ta) BAD_ARGS();
  napi_typedarray_type t; size_t len; void *data;
  NAPI_OK(napi_get_typedarray_info(env, argv[0], &t, &len, &data, NULL, NULL));
  if (t != napi_int32_array || len > 64) BAD_ARGS();
  pthread_mutex_lock(&g_lock);
  if (g_multi) { ntru_multi_destroy(g_multi); g_multi = NULL; }
  int rc = len ? ntru_multi_create((const int *)data, (int)len, &g_multi) : 0;
  pthread_mutex_unlock(&g_lock);
  if (rc) return throw_engine(env, rc);
  napi_value r; NAPI_OK(napi_create_int32(env, ntru_multi_engines(g_multi), &r)); return r;
}

static napi_value Supports(napi_env env, napi_callback_info info) {
  ARGS(2)
  int32_t N, mod;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &mod)) BAD_ARGS();
  napi_value r; NAPI_OK(napi_get_boolean(env, ntru_engine_supports(N, mod) != 0, &r)); return r;
}

/* setSamplerRounds(rounds) -> rounds in force: 20 (ChaCha20, RFC 8439: the default), 12 or 8 for the draw stream of sampleTernary /
 * sampleTernaryDev / pipelineBatch (ntru_engine_set_sampler_rounds); setSamplerRounds(0) only reads the setting. */
static napi_value SetSamplerRounds(napi_env env, napi_callback_info info) {
  ARGS(1)
  int32_t rounds;
  if (!get_i32(env, argv[0], &rounds)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  if (rounds != 0) {
    int rc = ntru_engine_set_sampler_rounds(g_engine, rounds);
    if (rc) return throw_engine(env, rc);
  }
  napi_value r; NAPI_OK(napi_create_int32(env, ntru_engine_get_sampler_rounds(g_engine), &r)); return r;
}

/* polymulSplit(N, mod, a:Uint16Array, b:Uint16Array, B, quot:Uint16Array, rem:Uint16Array) */
static napi_value PolymulSplit(napi_env env, napi_callback_info info) {
  ARGS(7)
  int32_t N, mod, B; void *a, *b, *quot, *rem;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &mod) || !get_i32(env, argv[4], &B) || N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[2], napi_uint16_array, n, 0, &a) || !get_buf(env, argv[3], napi_uint16_array, n, 0, &b) ||
      !get_buf(env, argv[5], napi_uint16_array, n, 0, &quot) || !get_buf(env, argv[6], napi_uint16_array, n, 0, &rem)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_polymul_split(g_engine, N, mod, a, b, B, quot, rem));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* splitByI(N, mod, a:Uint16Array[B*2N], B, quot, rem) */
static napi_value SplitByI(napi_env env, napi_callback_info info) {
  ARGS(6)
  int32_t N, mod, B; void *a, *quot, *rem;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &mod) || !get_i32(env, argv[3], &B) || N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[2], napi_uint16_array, 2 * n, 0, &a) || !get_buf(env, argv[4], napi_uint16_array, n, 0, &quot) ||
      !get_buf(env, argv[5], napi_uint16_array, n, 0, &rem)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_split_by_I(g_engine, N, mod, a, B, quot, rem));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* addBatch(N, mod, a, b, B, out) */
static napi_value AddBatch(napi_env env, napi_callback_info info) {
  ARGS(6)
  int32_t N, mod, B; void *a, *b, *out;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &mod) || !get_i32(env, argv[4], &B) || N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[2], napi_uint16_array, n, 0, &a) || !get_buf(env, argv[3], napi_uint16_array, n, 0, &b) ||
      !get_buf(env, argv[5], napi_uint16_array, n, 0, &out)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_add_batch(g_engine, N, mod, a, b, B, out));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* invertKeyBatch(N, q, p, f:Int8Array[B*N], B, fq:Uint16Array[B*N]|null, fp:Uint8Array[B*N]|null, flags:Uint8Array[B]) */
static napi_value InvertKeyBatch(napi_env env, napi_callback_info info) {
  ARGS(8)
  int32_t N, q, p, B; void *f, *fq, *fp, *flags;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[2], &p) || !get_i32(env, argv[4], &B) ||
      N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[3], napi_int8_array, n, 0, &f) || !get_buf(env, argv[5], napi_uint16_array, n, 1, &fq) ||
      !get_buf(env, argv[6], napi_uint8_array, n, 1, &fp) || !get_buf(env, argv[7], napi_uint8_array, (size_t)B, 0, &flags)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_invert_key_batch(g_engine, N, q, p, f, B, fq, fp, flags));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* publicKeyBatch(N, q, p, fq:Uint16Array[B*N], g:Int8Array[B*N], B, h:Uint16Array[B*N]) */
static napi_value PublicKeyBatch(napi_env env, napi_callback_info info) {
  ARGS(7)
  int32_t N, q, p, B; void *fq, *g, *h;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[2], &p) || !get_i32(env, argv[5], &B) ||
      N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[3], napi_uint16_array, n, 0, &fq) || !get_buf(env, argv[4], napi_int8_array, n, 0, &g) ||
      !get_buf(env, argv[6], napi_uint16_array, n, 0, &h)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_public_key_batch(g_engine, N, q, p, fq, g, B, h));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* encryptBatch(N, q, h:Uint16Array[N], r:Uint8Array[B*N], m:Uint8Array[B*N], B, e:Uint16Array, quotE:Uint16Array|null) */
static napi_value EncryptBatch(napi_env env, napi_callback_info info) {
  ARGS(8)
  int32_t N, q, B; void *h, *r, *m, *e, *quot;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[5], &B) || N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[2], napi_uint16_array, (size_t)N, 0, &h) || !get_buf(env, argv[3], napi_uint8_array, n, 0, &r) ||
      !get_buf(env, argv[4], napi_uint8_array, n, 0, &m) || !get_buf(env, argv[6], napi_uint16_array, n, 0, &e) ||
      !get_buf(env, argv[7], napi_uint16_array, n, 1, &quot)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, g_multi ? ntru_multi_encrypt_batch(g_multi, N, q, h, r, m, B, e, quot)
                   : ntru_encrypt_batch(g_engine, N, q, h, r, m, B, e, quot));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* decryptBatch(N, q, p, f:Int8Array[N], fp:Uint8Array[N], e:Uint16Array[B*N], B, value:Uint8Array,
 *              quot1:Uint16Array|null, rem1:Uint16Array|null, quot2:Uint8Array|null) */
static napi_value DecryptBatch(napi_env env, napi_callback_info info) {
  ARGS(11)
  int32_t N, q, p, B; void *f, *fp, *e, *value, *q1, *r1, *q2;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[2], &p) ||
      !get_i32(env, argv[6], &B) || N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[3], napi_int8_array, (size_t)N, 0, &f) || !get_buf(env, argv[4], napi_uint8_array, (size_t)N, 0, &fp) ||
      !get_buf(env, argv[5], napi_uint16_array, n, 0, &e) || !get_buf(env, argv[7], napi_uint8_array, n, 0, &value) ||
      !get_buf(env, argv[8], napi_uint16_array, n, 1, &q1) || !get_buf(env, argv[9], napi_uint16_array, n, 1, &r1) ||
      !get_buf(env, argv[10], napi_uint8_array, n, 1, &q2)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, g_multi ? ntru_multi_decrypt_batch(g_multi, N, q, p, f, fp, e, B, value, q1, r1, q2)
                   : ntru_decrypt_batch(g_engine, N, q, p, f, fp, e, B, value, q1, r1, q2));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* verifyKeysBatch(N, q, p, f:Int8Array, g:Int8Array, fq:Uint16Array, fp:Uint8Array, h:Uint16Array, B,
 *                 quotFq, remFq :Uint16Array, quotFp, remFp :Uint8Array, quotH, remH :Uint16Array, flags:Uint8Array[B]) */
static napi_value VerifyKeysBatch(napi_env env, napi_callback_info info) {
  ARGS(16)
  int32_t N, q, p, B; void *f, *g, *fq, *fp, *h, *o1, *o2, *o3, *o4, *o5, *o6, *fl;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[2], &p) ||
      !get_i32(env, argv[8], &B) || N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[3], napi_int8_array, n, 0, &f) || !get_buf(env, argv[4], napi_int8_array, n, 0, &g) ||
      !get_buf(env, argv[5], napi_uint16_array, n, 0, &fq) || !get_buf(env, argv[6], napi_uint8_array, n, 0, &fp) ||
      !get_buf(env, argv[7], napi_uint16_array, n, 0, &h) || !get_buf(env, argv[9], napi_uint16_array, n, 0, &o1) ||
      !get_buf(env, argv[10], napi_uint16_array, n, 0, &o2) || !get_buf(env, argv[11], napi_uint8_array, n, 0, &o3) ||
      !get_buf(env, argv[12], napi_uint8_array, n, 0, &o4) || !get_buf(env, argv[13], napi_uint16_array, n, 0, &o5) ||
      !get_buf(env, argv[14], napi_uint16_array, n, 0, &o6) || !get_buf(env, argv[15], napi_uint8_array, (size_t)B, 0, &fl)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, g_multi ? ntru_multi_verify_keys_batch(g_multi, N, q, p, f, g, fq, fp, h, B, o1, o2, o3, o4, o5, o6, fl)
                   : ntru_verify_keys_batch(g_engine, N, q, p, f, g, fq, fp, h, B, o1, o2, o3, o4, o5, o6, fl));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* sampleTernary(N, n1, n2, other, key:Uint32Array[8], firstItem:Number, B, out:Uint8Array[B*N]) */
static napi_value SampleTernary(napi_env env, napi_callback_info info) {
  ARGS(8)
  int32_t N, n1, n2, other, B; double first; void *key, *out;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &n1) || !get_i32(env, argv[2], &n2) ||
      !get_i32(env, argv[3], &other) || napi_get_value_double(env, argv[5], &first) != napi_ok ||
      !get_i32(env, argv[6], &B) || N < 1 || B < 0 || first < 0 || first > 9007199254740991.0) BAD_ARGS();
  if (!get_buf(env, argv[4], napi_uint32_array, 8, 0, &key) ||
      !get_buf(env, argv[7], napi_uint8_array, (size_t)N * (size_t)B, 0, &out)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_sample_ternary(g_engine, N, n1, n2, other, key, (uint64_t)first, B, out));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* packParams(maxVal, dataLen) -> [bits, perOutput, arrLen, outputSize] */
static napi_value PackParams(napi_env env, napi_callback_info info) {
  ARGS(2)
  int32_t mv, dl; int v[4];
  if (!get_i32(env, argv[0], &mv) || !get_i32(env, argv[1], &dl)) BAD_ARGS();
  int rc;
  ENGINE_CALL(rc, ntru_pack_params(mv, dl, &v[0], &v[1], &v[2], &v[3]));
  if (rc) return throw_engine(env, rc);
  napi_value arr; NAPI_OK(napi_create_array_with_length(env, 4, &arr));
  for (int i = 0; i < 4; i++) { napi_value n; NAPI_OK(napi_create_int32(env, v[i], &n)); NAPI_OK(napi_set_element(env, arr, i, n)); }
  return arr;
}

/* packBatch(maxVal, dataLen, data:Uint16Array[B*dataLen], B, out:BigUint64Array[B*outputSize*4]) */
static napi_value PackBatch(napi_env env, napi_callback_info info) {
  ARGS(5)
  int32_t mv, dl, B; void *data, *out; int v[4];
  if (!get_i32(env, argv[0], &mv) || !get_i32(env, argv[1], &dl) || !get_i32(env, argv[3], &B) || B < 0) BAD_ARGS();
  int rc;
  ENGINE_CALL(rc, ntru_pack_params(mv, dl, &v[0], &v[1], &v[2], &v[3]));
  if (rc) return throw_engine(env, rc);
  if (!get_buf(env, argv[2], napi_uint16_array, (size_t)dl * (size_t)B, 0, &data) ||
      !get_buf(env, argv[4], napi_biguint64_array, (size_t)v[3] * 4 * (size_t)B, 0, &out)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  ENGINE_CALL(rc, ntru_pack_batch(g_engine, mv, dl, data, B, out));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* unpackBatch(maxVal, packedBits, in:BigUint64Array[B*packedSize*4], packedSize, B, out:Uint16Array[B*packedSize*per]) */
static napi_value UnpackBatch(napi_env env, napi_callback_info info) {
  ARGS(6)
  int32_t mv, pb, ps, B; void *in, *out; int v[4];
  if (!get_i32(env, argv[0], &mv) || !get_i32(env, argv[1], &pb) || !get_i32(env, argv[3], &ps) ||
      !get_i32(env, argv[4], &B) || B < 0 || ps < 0) BAD_ARGS();
  int rc;
  ENGINE_CALL(rc, ntru_pack_params(mv, 0, &v[0], &v[1], &v[2], &v[3]));
  if (rc) return throw_engine(env, rc);
  const int per = pb / v[0];
  if (!get_buf(env, argv[2], napi_biguint64_array, (size_t)ps * 4 * (size_t)B, 0, &in) ||
      !get_buf(env, argv[5], napi_uint16_array, (size_t)ps * (size_t)(per > 0 ? per : 0) * (size_t)B, 0, &out)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  ENGINE_CALL(rc, ntru_unpack_batch(g_engine, mv, pb, in, ps, B, out));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* allocPinned(bytes) -> ArrayBuffer over page-locked memory (ntru_host_alloc): TypedArrays built on it are DMA'd in place
 * by the batch entry points instead of being staged.  Released when the ArrayBuffer is garbage collected. */
static void free_pinned(napi_env env, void *data, void *hint) { (void)env; (void)hint; ntru_host_free(data); }

static napi_value AllocPinned(napi_env env, napi_callback_info info) {
  ARGS(1)
  double bytes;
  if (napi_get_value_double(env, argv[0], &bytes) != napi_ok || bytes < 0 || bytes > 1e12) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  void *p = ntru_host_alloc((size_t)bytes);
  if (!p) return throw_engine(env, NTRU_ERR_HIP);
  napi_value ab;
  if (napi_create_external_arraybuffer(env, p, (size_t)bytes, free_pinned, NULL, &ab) != napi_ok) {
    ntru_host_free(p);
    napi_throw_error(env, NULL, "napi_create_external_arraybuffer failed");
    return NULL;
  }
  return ab;
}

/* genericOp(op, a:Float64Array, b:Float64Array, mod, out0:Float64Array, out1:Float64Array|null) -> [status, len0, len1]
 * op: 0 multiplyPolynomials, 1 dividePolynomials, 2 extendedEuclideanAlgorithm, 3 polyInv (one item; the generic family of
 * include/ntru_engine.h).  JS Numbers cross as doubles and are converted to the int64 the C ABI takes; out0 / out1 need
 * ntru_generic_capacity(a.length, b.length) elements (genericCapacity). */
static napi_value GenericCapacity(napi_env env, napi_callback_info info) {
  ARGS(2)
  int32_t la, lb;
  if (!get_i32(env, argv[0], &la) || !get_i32(env, argv[1], &lb)) BAD_ARGS();
  napi_value r; NAPI_OK(napi_create_int32(env, ntru_generic_capacity(la, lb), &r)); return r;
}

static napi_value GenericOp(napi_env env, napi_callback_info info) {
  ARGS(6)
  int32_t op; double mod; void *a = NULL, *b = NULL, *o0 = NULL, *o1 = NULL;
  size_t la = 0, lb = 0;
  napi_typedarray_type t; bool is_ta = false;
  if (!get_i32(env, argv[0], &op) || op < 0 || op > 3 || napi_get_value_double(env, argv[3], &mod) != napi_ok ||
      !(mod >= 1.0 && mod <= 9007199254740992.0) || mod != (double)(int64_t)mod) BAD_ARGS();
  if (napi_is_typedarray(env, argv[1], &is_ta) != napi_ok || !is_ta ||
      napi_get_typedarray_info(env, argv[1], &t, &la, &a, NULL, NULL) != napi_ok || t != napi_float64_array) BAD_ARGS();
  if (napi_is_typedarray(env, argv[2], &is_ta) != napi_ok || !is_ta ||
      napi_get_typedarray_info(env, argv[2], &t, &lb, &b, NULL, NULL) != napi_ok || t != napi_float64_array) BAD_ARGS();
  const size_t cap = (size_t)ntru_generic_capacity((int)la, (int)lb);
  if (!get_buf(env, argv[4], napi_float64_array, cap, 0, &o0) || !get_buf(env, argv[5], napi_float64_array, cap, 1, &o1)) BAD_ARGS();
  if ((op == 1 || op == 2) && !o1) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int64_t *buf = (int64_t *)malloc((la + lb + 2 * cap + 2) * sizeof(int64_t));
  if (!buf) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  int64_t *ia = buf, *ib = ia + la + 1, *r0 = ib + lb + 1, *r1 = r0 + cap;
  for (size_t i = 0; i < la + lb; i++) {                      /* doubles -> int64: integers only, and only where the cast is defined */
    const double v = i < la ? ((double *)a)[i] : ((double *)b)[i - la];
    if (!(v >= -9007199254740992.0 && v <= 9007199254740992.0) || v != (double)(int64_t)v) {
      free(buf);
      napi_throw_type_error(env, NULL, "coefficients must be integers");
      return NULL;
    }
    if (i < la) ia[i] = (int64_t)v; else ib[i - la] = (int64_t)v;
  }
  int32_t len0 = 0, len1 = 0; uint8_t st = 0;
  int rc;
  if (op == 0) ENGINE_CALL(rc, ntru_generic_multiply(g_engine, (int)la, (int)lb, (int64_t)mod, ia, ib, 1, r0, &len0));
  else if (op == 1) ENGINE_CALL(rc, ntru_generic_divide(g_engine, (int)la, (int)lb, (int64_t)mod, ia, ib, 1, r0, &len0, r1, &len1, &st));
  else if (op == 2) ENGINE_CALL(rc, ntru_generic_eea(g_engine, (int)la, (int)lb, (int64_t)mod, ia, ib, 1, r0, &len0, r1, &len1, &st));
  else ENGINE_CALL(rc, ntru_generic_poly_inv(g_engine, (int)la, (int)lb, (int64_t)mod, ia, ib, 1, r0, &len0, &st));
  if (!rc) {
    for (int32_t i = 0; i < len0; i++) ((double *)o0)[i] = (double)r0[i];
    if (o1) for (int32_t i = 0; i < len1; i++) ((double *)o1)[i] = (double)r1[i];
  }
  free(buf);
  if (rc) return throw_engine(env, rc);
  napi_value arr; NAPI_OK(napi_create_array_with_length(env, 3, &arr));
  const int32_t v[3] = {st, len0, len1};
  for (int i = 0; i < 3; i++) { napi_value n; NAPI_OK(napi_create_int32(env, v[i], &n)); NAPI_OK(napi_set_element(env, arr, i, n)); }
  return arr;
}

/* ---- device-resident use (additive).  pipelineBatch chains sampler -> encryptBits -> decryptBits -> packOutput on the GPU for a batch
 *      of host plaintexts (ntru_pipeline_batch); devAlloc / devUpload / devDownload / devFree plus the *Dev calls expose the engine's
 *      *_dev entry points on opaque device-buffer handles, for callers that want to compose the stages themselves.  Every *Dev call
 *      checks the byte size of each handle against what the kernel will touch before anything is launched. */

/* pipelineBatch(N, q, p, h:Uint16Array[N], f:Int8Array[N]|null, fp:Uint8Array[N]|null, key:Uint32Array[8]|null, firstItem, n1, n2,
 *               r:Uint8Array[B*N]|null, m:Uint8Array[B*N], B, rOut:Uint8Array|null, e:Uint16Array|null, value:Uint8Array|null,
 *               packed:BigUint64Array[B*outputSize*4]|null) */
static napi_value PipelineBatch(napi_env env, napi_callback_info info) {
  ARGS(17)
  int32_t N, q, p, n1, n2, B; double first; void *h, *f, *fp, *key, *r, *m, *r_out, *e, *value, *packed;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[2], &p) ||
      napi_get_value_double(env, argv[7], &first) != napi_ok || !get_i32(env, argv[8], &n1) || !get_i32(env, argv[9], &n2) ||
      !get_i32(env, argv[12], &B) || N < 1 || B < 0 || first < 0 || first > 9007199254740991.0) BAD_ARGS();
  const size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[3], napi_uint16_array, (size_t)N, 0, &h) || !get_buf(env, argv[4], napi_int8_array, (size_t)N, 1, &f) ||
      !get_buf(env, argv[5], napi_uint8_array, (size_t)N, 1, &fp) || !get_buf(env, argv[6], napi_uint32_array, 8, 1, &key) ||
      !get_buf(env, argv[10], napi_uint8_array, n, 1, &r) || !get_buf(env, argv[11], napi_uint8_array, n, 0, &m) ||
      !get_buf(env, argv[13], napi_uint8_array, n, 1, &r_out) || !get_buf(env, argv[14], napi_uint16_array, n, 1, &e) ||
      !get_buf(env, argv[15], napi_uint8_array, n, 1, &value)) BAD_ARGS();
  size_t need_packed = 0;
  {
    int bits, per, al, os;
    if (ntru_pack_params(f ? p - 1 : q - 1, N, &bits, &per, &al, &os) == 0) need_packed = (size_t)B * (size_t)os * 4;
  }
  if (!get_buf(env, argv[16], napi_biguint64_array, need_packed, 1, &packed)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_pipeline_batch(g_engine, N, q, p, h, f, fp, key, (uint64_t)first, n1, n2, r, m, B, r_out, e, value, packed));
  return rc ? throw_engine(env, rc) : undefined(env);
}

typedef struct DevBuf { void *p; size_t bytes; struct DevBuf *next; } DevBuf;
/* Every live handle of THIS addon.  An external value made by anybody else (another addon, a foreign napi_external) carries a data
 * pointer that is not in this list and is refused by get_dev without ever being dereferenced. */
static DevBuf *g_devs = NULL;
static pthread_mutex_t g_devs_lock = PTHREAD_MUTEX_INITIALIZER;
static void devs_add(DevBuf *b) { pthread_mutex_lock(&g_devs_lock); b->next = g_devs; g_devs = b; pthread_mutex_unlock(&g_devs_lock); }
static void devs_remove(DevBuf *b) {
  pthread_mutex_lock(&g_devs_lock);
  for (DevBuf **pp = &g_devs; *pp; pp = &(*pp)->next) if (*pp == b) { *pp = b->next; break; }
  pthread_mutex_unlock(&g_devs_lock);
}
static int devs_has(const void *data) {
  int found = 0;
  pthread_mutex_lock(&g_devs_lock);
  for (DevBuf *b = g_devs; b; b = b->next) if ((const void *)b == data) { found = 1; break; }
  pthread_mutex_unlock(&g_devs_lock);
  return found;
}

static void devbuf_finalize(napi_env env, void *data, void *hint) {
  (void)env; (void)hint;
  DevBuf *b = (DevBuf *)data;
  devs_remove(b);
  if (b->p) {                       /* not freed explicitly: release it with the handle (the engine may be gone already: then leak) */
    pthread_mutex_lock(&g_lock);
    if (g_engine) (void)ntru_dev_free(g_engine, b->p);
    pthread_mutex_unlock(&g_lock);
  }
  free(b);
}

static DevBuf *get_dev(napi_env env, napi_value v, size_t need, int optional, int *ok) {
  napi_valuetype vt;
  *ok = 0;
  if (napi_typeof(env, v, &vt) != napi_ok) return NULL;
  if (vt == napi_null || vt == napi_undefined) { *ok = optional; return NULL; }
  if (vt != napi_external) return NULL;
  void *data = NULL;
  if (napi_get_value_external(env, v, &data) != napi_ok || !data || !devs_has(data)) return NULL;   /* not one of ours: refused */
  DevBuf *b = (DevBuf *)data;
  if (!b->p || b->bytes < need) return NULL;
  *ok = 1;
  return b;
}

/* devAlloc(bytes) -> handle */
static napi_value DevAlloc(napi_env env, napi_callback_info info) {
  ARGS(1)
  double bytes;
  if (napi_get_value_double(env, argv[0], &bytes) != napi_ok || bytes < 0 || bytes > 281474976710656.0) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  DevBuf *b = (DevBuf *)calloc(1, sizeof *b);
  if (!b) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  int rc;
  ENGINE_CALL(rc, ntru_dev_alloc(g_engine, (size_t)bytes, &b->p));
  if (rc) { free(b); return throw_engine(env, rc); }
  b->bytes = (size_t)bytes;
  devs_add(b);
  napi_value ext;
  if (napi_create_external(env, b, devbuf_finalize, NULL, &ext) != napi_ok) {
    devs_remove(b);
    ENGINE_CALL(rc, ntru_dev_free(g_engine, b->p)); free(b);
    napi_throw_error(env, NULL, "napi_create_external failed"); return NULL;
  }
  return ext;
}

/* devFree(handle) */
static napi_value DevFree(napi_env env, napi_callback_info info) {
  ARGS(1)
  int ok; DevBuf *b = get_dev(env, argv[0], 0, 0, &ok);
  if (!ok) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_dev_free(g_engine, b->p));
  b->p = NULL; b->bytes = 0;
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* The bytes of any TypedArray. */
static int get_any(napi_env env, napi_value v, void **data, size_t *bytes) {
  bool is_ta = false;
  if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) return 0;
  napi_typedarray_type t; size_t len;
  if (napi_get_typedarray_info(env, v, &t, &len, data, NULL, NULL) != napi_ok) return 0;
  static const size_t w[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
  if ((int)t < 0 || (size_t)t >= sizeof w / sizeof w[0]) return 0;
  *bytes = len * w[t];
  return 1;
}

/* devUpload(handle, src:TypedArray): the whole array to the start of the buffer */
static napi_value DevUpload(napi_env env, napi_callback_info info) {
  ARGS(2)
  void *src; size_t bytes;
  if (!get_any(env, argv[1], &src, &bytes)) BAD_ARGS();
  int ok; DevBuf *b = get_dev(env, argv[0], bytes, 0, &ok);
  if (!ok) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_dev_upload(g_engine, b->p, src, bytes));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* devDownload(dst:TypedArray, handle): the first dst.byteLength bytes of the buffer; waits for the engine's stream */
static napi_value DevDownload(napi_env env, napi_callback_info info) {
  ARGS(2)
  void *dst; size_t bytes;
  if (!get_any(env, argv[0], &dst, &bytes)) BAD_ARGS();
  int ok; DevBuf *b = get_dev(env, argv[1], bytes, 0, &ok);
  if (!ok) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_dev_download(g_engine, dst, b->p, bytes));
  return rc ? throw_engine(env, rc) : undefined(env);
}

#define DEV(i, need, opt, var) DevBuf *var = get_dev(env, argv[i], (need), (opt), &ok); if (!ok) BAD_ARGS();

/* sampleTernaryDev(N, n1, n2, other, key:Uint32Array[8], firstItem, B, out:handle[B*N]) */
static napi_value SampleTernaryDev(napi_env env, napi_callback_info info) {
  ARGS(8)
  int32_t N, n1, n2, other, B; double first; void *key; int ok;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &n1) || !get_i32(env, argv[2], &n2) || !get_i32(env, argv[3], &other) ||
      napi_get_value_double(env, argv[5], &first) != napi_ok || !get_i32(env, argv[6], &B) || N < 1 || B < 0 || first < 0 ||
      first > 9007199254740991.0 || !get_buf(env, argv[4], napi_uint32_array, 8, 0, &key)) BAD_ARGS();
  DEV(7, (size_t)N * (size_t)B, 0, out)
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_sample_ternary_dev(g_engine, N, n1, n2, other, key, (uint64_t)first, B, out->p));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* encryptBatchDev(N, q, h:handle[N u16], r:handle[B*N u8], m:handle[B*N u8], B, e:handle[B*N u16], quotE:handle|null) */
static napi_value EncryptBatchDev(napi_env env, napi_callback_info info) {
  ARGS(8)
  int32_t N, q, B; int ok;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[5], &B) || N < 1 || B < 0) BAD_ARGS();
  const size_t n = (size_t)N * (size_t)B;
  DEV(2, 2 * (size_t)N, 0, h) DEV(3, n, 0, r) DEV(4, n, 0, m) DEV(6, 2 * n, 0, e) DEV(7, 2 * n, 1, quot)
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_encrypt_batch_dev(g_engine, N, q, h->p, r->p, m->p, B, e->p, quot ? quot->p : NULL));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* decryptBatchDev(N, q, p, f:handle[N i8], fp:handle[N u8], e:handle[B*N u16], B, value:handle[B*N u8],
 *                 quot1:handle|null, rem1:handle|null, quot2:handle|null) */
static napi_value DecryptBatchDev(napi_env env, napi_callback_info info) {
  ARGS(11)
  int32_t N, q, p, B; int ok;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[2], &p) || !get_i32(env, argv[6], &B) ||
      N < 1 || B < 0) BAD_ARGS();
  const size_t n = (size_t)N * (size_t)B;
  DEV(3, (size_t)N, 0, f) DEV(4, (size_t)N, 0, fp) DEV(5, 2 * n, 0, e) DEV(7, n, 0, value) DEV(8, 2 * n, 1, q1) DEV(9, 2 * n, 1, r1) DEV(10, n, 1, q2)
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, ntru_decrypt_batch_dev(g_engine, N, q, p, f->p, fp->p, e->p, B, value->p, q1 ? q1->p : NULL, r1 ? r1->p : NULL,
                                         q2 ? q2->p : NULL));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* packBatchDev(maxVal, dataLen, data:handle, B, out:handle[B*outputSize*32 bytes], bytes:boolean)   bytes: data holds uint8 values */
static napi_value PackBatchDev(napi_env env, napi_callback_info info) {
  ARGS(6)
  int32_t max_val, data_len, B; bool is_bytes; int ok;
  if (!get_i32(env, argv[0], &max_val) || !get_i32(env, argv[1], &data_len) || !get_i32(env, argv[3], &B) || data_len < 0 || B < 0 ||
      napi_get_value_bool(env, argv[5], &is_bytes) != napi_ok) BAD_ARGS();
  int bits, per, al, os;
  if (ntru_pack_params(max_val, data_len, &bits, &per, &al, &os)) return throw_engine(env, NTRU_ERR_ARG);
  DEV(2, (size_t)B * (size_t)data_len * (is_bytes ? 1 : 2), 0, data) DEV(4, (size_t)B * (size_t)os * 32, 0, out)
  if (!ensure_engine(env)) return NULL;
  int rc;
  ENGINE_CALL(rc, is_bytes ? ntru_pack_bytes_batch_dev(g_engine, max_val, data_len, data->p, B, out->p)
                           : ntru_pack_batch_dev(g_engine, max_val, data_len, data->p, B, out->p));
  return rc ? throw_engine(env, rc) : undefined(env);
}

/* ---- asynchronous batch calls (additive; the reference API stays synchronous).  encryptBatchAsync / decryptBatchAsync take
 *      the arguments of their synchronous twins and return a Promise; the engine call runs on a libuv worker thread, so the
 *      event loop keeps turning while a 2^18-item batch (tens of milliseconds of PCIe) is in flight.  The typed arrays are
 *      pinned by references until the Promise settles; the caller must not touch the output arrays before that. */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref keep[12];
  int n_keep;
  int kind;                       /* 0 encrypt, 1 decrypt, 2 pipeline */
  int N, q, p, B, n1, n2;
  uint64_t first;
  void *ptr[12];
  int rc;
  char err[400];
} AsyncJob;

static void async_execute(napi_env env, void *data) {
  (void)env;
  AsyncJob *j = (AsyncJob *)data;
  pthread_mutex_lock(&g_lock);
  if (!g_engine) { j->rc = NTRU_ERR_ARG; snprintf(j->err, sizeof j->err, "ntru engine not created"); }
  else {
    if (j->kind == 0)
      j->rc = g_multi ? ntru_multi_encrypt_batch(g_multi, j->N, j->q, j->ptr[0], j->ptr[1], j->ptr[2], j->B, j->ptr[3], j->ptr[4])
                      : ntru_encrypt_batch(g_engine, j->N, j->q, j->ptr[0], j->ptr[1], j->ptr[2], j->B, j->ptr[3], j->ptr[4]);
    else if (j->kind == 2)
      j->rc = ntru_pipeline_batch(g_engine, j->N, j->q, j->p, j->ptr[0], j->ptr[1], j->ptr[2], j->ptr[3], j->first, j->n1, j->n2, j->ptr[4],
                                  j->ptr[5], j->B, j->ptr[6], j->ptr[7], j->ptr[8], j->ptr[9]);
    else
      j->rc = g_multi ? ntru_multi_decrypt_batch(g_multi, j->N, j->q, j->p, j->ptr[0], j->ptr[1], j->ptr[2], j->B, j->ptr[3],
                                                 j->ptr[4], j->ptr[5], j->ptr[6])
                      : ntru_decrypt_batch(g_engine, j->N, j->q, j->p, j->ptr[0], j->ptr[1], j->ptr[2], j->B, j->ptr[3], j->ptr[4],
                                           j->ptr[5], j->ptr[6]);
    if (j->rc) snprintf(j->err, sizeof j->err, "ntru engine error %d: %s", j->rc, ntru_last_error());   /* per thread: read it here */
  }
  pthread_mutex_unlock(&g_lock);
}

static void async_complete(napi_env env, napi_status status, void *data) {
  AsyncJob *j = (AsyncJob *)data;
  napi_value v;
  if (status == napi_ok && j->rc == 0) {
    napi_get_undefined(env, &v);
    napi_resolve_deferred(env, j->deferred, v);
  } else {
    napi_value msg;
    napi_create_string_utf8(env, status == napi_ok ? j->err : "ntru engine: asynchronous work was cancelled", NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &v);
    napi_reject_deferred(env, j->deferred, v);
  }
  for (int i = 0; i < j->n_keep; i++) napi_delete_reference(env, j->keep[i]);
  napi_delete_async_work(env, j->work);
  free(j);
}

static napi_value async_start(napi_env env, AsyncJob *j, napi_value *hold, int n_hold, const char *name) {
  napi_value promise, rname;
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) { free(j); napi_throw_error(env, NULL, "napi_create_promise failed"); return NULL; }
  for (int i = 0; i < n_hold; i++) {
    napi_valuetype vt;
    if (napi_typeof(env, hold[i], &vt) == napi_ok && vt == napi_object &&
        napi_create_reference(env, hold[i], 1, &j->keep[j->n_keep]) == napi_ok) j->n_keep++;
  }
  napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &rname);
  if (napi_create_async_work(env, NULL, rname, async_execute, async_complete, j, &j->work) != napi_ok ||
      napi_queue_async_work(env, j->work) != napi_ok) {
    for (int i = 0; i < j->n_keep; i++) napi_delete_reference(env, j->keep[i]);
    free(j);
    napi_throw_error(env, NULL, "could not queue asynchronous work");
    return NULL;
  }
  return promise;
}

/* encryptBatchAsync(N, q, h, r, m, B, e, quotE|null) -> Promise<undefined> */
static napi_value EncryptBatchAsync(napi_env env, napi_callback_info info) {
  ARGS(8)
  int32_t N, q, B; void *h, *r, *m, *e, *quot;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[5], &B) || N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[2], napi_uint16_array, (size_t)N, 0, &h) || !get_buf(env, argv[3], napi_uint8_array, n, 0, &r) ||
      !get_buf(env, argv[4], napi_uint8_array, n, 0, &m) || !get_buf(env, argv[6], napi_uint16_array, n, 0, &e) ||
      !get_buf(env, argv[7], napi_uint16_array, n, 1, &quot)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  AsyncJob *j = (AsyncJob *)calloc(1, sizeof *j);
  if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  j->kind = 0; j->N = N; j->q = q; j->B = B;
  j->ptr[0] = h; j->ptr[1] = r; j->ptr[2] = m; j->ptr[3] = e; j->ptr[4] = quot;
  napi_value hold[5] = {argv[2], argv[3], argv[4], argv[6], argv[7]};
  return async_start(env, j, hold, 5, "ntru.encryptBatchAsync");
}

/* decryptBatchAsync(N, q, p, f, fp, e, B, value, quot1|null, rem1|null, quot2|null) -> Promise<undefined> */
static napi_value DecryptBatchAsync(napi_env env, napi_callback_info info) {
  ARGS(11)
  int32_t N, q, p, B; void *f, *fp, *e, *value, *q1, *r1, *q2;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[2], &p) ||
      !get_i32(env, argv[6], &B) || N < 1 || B < 0) BAD_ARGS();
  size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[3], napi_int8_array, (size_t)N, 0, &f) || !get_buf(env, argv[4], napi_uint8_array, (size_t)N, 0, &fp) ||
      !get_buf(env, argv[5], napi_uint16_array, n, 0, &e) || !get_buf(env, argv[7], napi_uint8_array, n, 0, &value) ||
      !get_buf(env, argv[8], napi_uint16_array, n, 1, &q1) || !get_buf(env, argv[9], napi_uint16_array, n, 1, &r1) ||
      !get_buf(env, argv[10], napi_uint8_array, n, 1, &q2)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  AsyncJob *j = (AsyncJob *)calloc(1, sizeof *j);
  if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  j->kind = 1; j->N = N; j->q = q; j->p = p; j->B = B;
  j->ptr[0] = f; j->ptr[1] = fp; j->ptr[2] = e; j->ptr[3] = value; j->ptr[4] = q1; j->ptr[5] = r1; j->ptr[6] = q2;
  napi_value hold[7] = {argv[3], argv[4], argv[5], argv[7], argv[8], argv[9], argv[10]};
  return async_start(env, j, hold, 7, "ntru.decryptBatchAsync");
}

/* pipelineBatchAsync(...the arguments of pipelineBatch...) -> Promise<undefined> */
static napi_value PipelineBatchAsync(napi_env env, napi_callback_info info) {
  ARGS(17)
  int32_t N, q, p, n1, n2, B; double first; void *h, *f, *fp, *key, *r, *m, *r_out, *e, *value, *packed;
  if (!get_i32(env, argv[0], &N) || !get_i32(env, argv[1], &q) || !get_i32(env, argv[2], &p) ||
      napi_get_value_double(env, argv[7], &first) != napi_ok || !get_i32(env, argv[8], &n1) || !get_i32(env, argv[9], &n2) ||
      !get_i32(env, argv[12], &B) || N < 1 || B < 0 || first < 0 || first > 9007199254740991.0) BAD_ARGS();
  const size_t n = (size_t)N * (size_t)B;
  if (!get_buf(env, argv[3], napi_uint16_array, (size_t)N, 0, &h) || !get_buf(env, argv[4], napi_int8_array, (size_t)N, 1, &f) ||
      !get_buf(env, argv[5], napi_uint8_array, (size_t)N, 1, &fp) || !get_buf(env, argv[6], napi_uint32_array, 8, 1, &key) ||
      !get_buf(env, argv[10], napi_uint8_array, n, 1, &r) || !get_buf(env, argv[11], napi_uint8_array, n, 0, &m) ||
      !get_buf(env, argv[13], napi_uint8_array, n, 1, &r_out) || !get_buf(env, argv[14], napi_uint16_array, n, 1, &e) ||
      !get_buf(env, argv[15], napi_uint8_array, n, 1, &value)) BAD_ARGS();
  size_t need_packed = 0;
  {
    int bits, per, al, os;
    if (ntru_pack_params(f ? p - 1 : q - 1, N, &bits, &per, &al, &os) == 0) need_packed = (size_t)B * (size_t)os * 4;
  }
  if (!get_buf(env, argv[16], napi_biguint64_array, need_packed, 1, &packed)) BAD_ARGS();
  if (!ensure_engine(env)) return NULL;
  AsyncJob *j = (AsyncJob *)calloc(1, sizeof *j);
  if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
  j->kind = 2; j->N = N; j->q = q; j->p = p; j->B = B; j->n1 = n1; j->n2 = n2; j->first = (uint64_t)first;
  void *ptrs[10] = {h, f, fp, key, r, m, r_out, e, value, packed};
  memcpy(j->ptr, ptrs, sizeof ptrs);
  napi_value hold[10] = {argv[3], argv[4], argv[5], argv[6], argv[10], argv[11], argv[13], argv[14], argv[15], argv[16]};
  return async_start(env, j, hold, 10, "ntru.pipelineAsync");
}

static napi_value Init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
    {"deviceCount", NULL, DeviceCount, NULL, NULL, NULL, napi_default, NULL},
    {"create", NULL, Create, NULL, NULL, NULL, napi_default, NULL},
    {"destroy", NULL, Destroy, NULL, NULL, NULL, napi_default, NULL},
    {"useDevices", NULL, UseDevices, NULL, NULL, NULL, napi_default, NULL},
    {"supports", NULL, Supports, NULL, NULL, NULL, napi_default, NULL},
    {"setSamplerRounds", NULL, SetSamplerRounds, NULL, NULL, NULL, napi_default, NULL},
    {"polymulSplit", NULL, PolymulSplit, NULL, NULL, NULL, napi_default, NULL},
    {"splitByI", NULL, SplitByI, NULL, NULL, NULL, napi_default, NULL},
    {"addBatch", NULL, AddBatch, NULL, NULL, NULL, napi_default, NULL},
    {"encryptBatch", NULL, EncryptBatch, NULL, NULL, NULL, napi_default, NULL},
    {"decryptBatch", NULL, DecryptBatch, NULL, NULL, NULL, napi_default, NULL},
    {"verifyKeysBatch", NULL, VerifyKeysBatch, NULL, NULL, NULL, napi_default, NULL},
    {"publicKeyBatch", NULL, PublicKeyBatch, NULL, NULL, NULL, napi_default, NULL},
    {"invertKeyBatch", NULL, InvertKeyBatch, NULL, NULL, NULL, napi_default, NULL},
    {"sampleTernary", NULL, SampleTernary, NULL, NULL, NULL, napi_default, NULL},
    {"packParams", NULL, PackParams, NULL, NULL, NULL, napi_default, NULL},
    {"packBatch", NULL, PackBatch, NULL, NULL, NULL, napi_default, NULL},
    {"unpackBatch", NULL, UnpackBatch, NULL, NULL, NULL, napi_default, NULL},
    {"allocPinned", NULL, AllocPinned, NULL, NULL, NULL, napi_default, NULL},
    {"genericCapacity", NULL, GenericCapacity, NULL, NULL, NULL, napi_default, NULL},
    {"genericOp", NULL, GenericOp, NULL, NULL, NULL, napi_default, NULL},
    {"pipelineBatch", NULL, PipelineBatch, NULL, NULL, NULL, napi_default, NULL},
    {"pipelineBatchAsync", NULL, PipelineBatchAsync, NULL, NULL, NULL, napi_default, NULL},
    {"devAlloc", NULL, DevAlloc, NULL, NULL, NULL, napi_default, NULL},
    {"devFree", NULL, DevFree, NULL, NULL, NULL, napi_default, NULL},
    {"devUpload", NULL, DevUpload, NULL, NULL, NULL, napi_default, NULL},
    {"devDownload", NULL, DevDownload, NULL, NULL, NULL, napi_default, NULL},
    {"sampleTernaryDev", NULL, SampleTernaryDev, NULL, NULL, NULL, napi_default, NULL},
    {"encryptBatchDev", NULL, EncryptBatchDev, NULL, NULL, NULL, napi_default, NULL},
    {"decryptBatchDev", NULL, DecryptBatchDev, NULL, NULL, NULL, napi_default, NULL},
    {"packBatchDev", NULL, PackBatchDev, NULL, NULL, NULL, napi_default, NULL},
    {"encryptBatchAsync", NULL, EncryptBatchAsync, NULL, NULL, NULL, napi_default, NULL},
    {"decryptBatchAsync", NULL, DecryptBatchAsync, NULL, NULL, NULL, napi_default, NULL},
  };
  if (napi_define_properties(env, exports, sizeof props / sizeof props[0], props) != napi_ok) return NULL;
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
