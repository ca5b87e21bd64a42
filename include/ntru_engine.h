/*
 * ntru_engine.h -- C ABI of the MI355X NTRU polynomial-ring engine.
 *
 * This is the drop-in boundary for the hot path of numtel/ntru-circom (reference file index.js):
 * the linear product (multiplyPolynomials, index.js:319-355), the quotient/remainder split by
 * I = 1 - x^N (dividePolynomials with b = I, index.js:358-401), the coefficient-wise reductions
 * (addPolynomials index.js:235-244, centred lift index.js:117) and the three scheme methods built
 * from them (encryptBits :87-110, decryptBits :111-140, verifyKeysInputs :141-197).
 *
 * The reference has no FFI of its own (it is a single ES module); the bindings a maintainer adds
 * are shown in INTEGRATION.md (N-API addon in ntru-circom_amd/js/, ctypes in ntru-circom_amd/engine.py).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, caller-owned buffers, nothing allocated across the ABI.
 *   - every batch array is row-major [B][N] with a fixed stride of N elements, zero padded: row b holds
 *     the N coefficients of item b.  The reference's witness arrays of length N+1 always end in 0
 *     (index.js:101-102,127-130,174-175): that element and the trimming of `value`
 *     (trimPolynomial, index.js:218-221) are added by the host-side shim, not stored on the device.
 *   - mod-q coefficients are uint16_t (q a power of two, 2 <= q <= 65536); ternary / mod-p coefficients
 *     are uint8_t; signed ternary key material f, g is int8_t in {-1, 0, 1}.
 *   - every function returns 0 on success or one of the NTRU_ERR_* codes; ntru_last_error() then
 *     returns a message for the calling thread.
 *   - *_dev entry points take DEVICE pointers and enqueue work on the engine's stream without
 *     synchronising; the same names without _dev take HOST pointers and return when the results are in
 *     the caller's buffers: the batch is cut into chunks that flow through two engine-owned streams
 *     (H2D, kernel and D2H of neighbouring chunks overlap), through pinned staging arenas that only grow
 *     -- or with no staging at all for buffers from ntru_host_alloc.  An engine is used by one thread at a time.
 *   - there is no CPU fallback: without a usable HIP device ntru_engine_create fails.
 *   - symbol preconditions (what the reference itself always produces): r in {0,1,2}; f, g in {-1,0,1}; fp and the
 *     plaintext-side values below p.  The fast kernels rely on them (the ternary add path steps over these operands by
 *     symbol class, the matrix-core path carries them as int8 digits: r <= 3, |f| <= 1); any other value is a caller
 *     error with unspecified results.  ntru_engine_set_kernel_path(1) selects the multiply-accumulate kernels, which
 *     accept arbitrary operand values for r and h, e, fq.
 *   - shared-key encrypt / decrypt with 64 <= N <= 1024 and q <= 8192 run on the int8 matrix cores (batch x Toeplitz
 *     matrix of the key, exact), and so does verify_keys with 128 <= N <= 1024, q <= 8192, p == 3 (each per-item product
 *     as a 32-row matrix product per tile distance), polymul_split with a power-of-two modulus <= 8192, the public key
 *     and the Newton rounds of the key inversion in the same N range; everything else on the vector-ALU kernel families.
 *     Pointers may have any alignment.
 */
#ifndef NTRU_ENGINE_H
#define NTRU_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTRU_OK 0
#define NTRU_ERR_NO_DEVICE 1   /* no HIP device / device id out of range */
#define NTRU_ERR_ARG 2         /* NULL pointer, negative size, ...       */
#define NTRU_ERR_UNSUPPORTED 3 /* parameter set outside what the kernels implement */
#define NTRU_ERR_HIP 4         /* a HIP runtime call failed               */

/* bits of the per-item flags byte written by ntru_verify_keys_batch: set when the reference's
 * verifyKeysInputs would throw 'invalid fq' / 'invalid fp' / 'invalid h' (index.js:159-166). */
#define NTRU_FLAG_INVALID_FQ 1
#define NTRU_FLAG_INVALID_FP 2
#define NTRU_FLAG_INVALID_H 4
/* bits of the per-key flags byte written by ntru_invert_key_batch */
#define NTRU_FLAG_NOT_UNIT_MOD2 8  /* f has no inverse modulo 2 (hence none modulo q): fq is zero */
#define NTRU_FLAG_NOT_UNIT_MODP 16 /* f has no inverse modulo p: fp is zero */

typedef struct ntru_engine ntru_engine_t;

/* Number of HIP devices visible to the process (0 if none / runtime unusable). */
int ntru_engine_device_count(void);

/* Create an engine bound to HIP device `device`.  Work is enqueued on the NULL stream until
 * ntru_engine_set_stream is called. */
int ntru_engine_create(int device, ntru_engine_t **out);
void ntru_engine_destroy(ntru_engine_t *eng);

/* Use `hip_stream` (a hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) for all later calls.
 * The engine never owns the stream. */
int ntru_engine_set_stream(ntru_engine_t *eng, void *hip_stream);
/* Tuning / test knob: 0 = pick the fastest applicable kernel family (default), 1 = always the packed-u16 MAC
 * kernels, 2 = the ternary add path wherever it applies, 3 = the add path without its dot8 product, 4 = the int8
 * matrix-core path wherever it applies (encrypt / decrypt: shared key, q <= 8192, N <= 1024; verify_keys: q <= 8192,
 * p == 3, 64 <= N <= 1024), even for small N, as two independent four-wave workgroups per CU (k_encrypt_m, k_decrypt_m),
 * 5 = the matrix-core path as auto-selection runs it at large N, for every N: decrypt as ONE workgroup of two four-wave groups
 * whose matrix-loop and epilogue phases are interleaved by barriers (k_decrypt_m8; k_decrypt_m where 160 KB of LDS do not hold
 * two groups), encrypt with direct-to-LDS operand loads (k_encrypt_md; k_encrypt_m for rows that do not fit one such
 * instruction).  0 picks k_encrypt_md and, for N > 512 with every witness array, k_decrypt_m8.  Results are identical on every
 * path.  Paths 6-12 (role-split / chunked-store / lock-step / row-image encrypt, direct-to-LDS decrypt, decrypt with an fp4 second product,
 * verify_keys on the 16-row matrix tile: built, bit-exact, measured slower or equal; the timings of paths 11 and 12 include scratch spills
 * outside their loops)
 * exist only in a library built with -DNTRU_EXPERIMENTS (`make -C ntru-circom_amd/csrc experiments`); this call refuses them
 * otherwise.
 * Streams: the *_dev calls that need temporaries (key inversion, the generic family) share one engine-owned scratch buffer.
 * Calls on one stream are ordered by the stream; after ntru_engine_set_stream a call first makes the new stream wait (on the
 * device) for the previous stream's use of that buffer.  The buffer only grows, and growing it frees the old one with hipFree,
 * which waits for the whole device. */
int ntru_engine_set_kernel_path(ntru_engine_t *eng, int path);
/* Name of the kernel the last *_dev call on this engine launched, e.g. "k_decrypt_s<13,13>" (for reports). */
const char *ntru_engine_last_kernel(ntru_engine_t *eng);
/* Block until everything enqueued on the engine's stream has finished. */
int ntru_engine_synchronize(ntru_engine_t *eng);

/* Message describing the last error on this thread ("" if none). */
const char *ntru_last_error(void);

/* ---- device-resident use from a host language without HIP of its own (the N-API addon) --------------------------------------
 * ntru_pipeline_batch chains generateCustomArray (index.js:461-488: the r of encryptBits, drawn on the device from a ChaCha20
 * stream) -> encryptBits (index.js:87-110) -> decryptBits (index.js:111-140) -> packOutput (index.js:572-596) for a batch of HOST
 * plaintexts m [B][N]; the intermediates stay on the GPU, chunk by chunk through the engine's three stage streams (upload, compute, download).
 *   key != NULL: r is sampled (n1 ones, n2 entries p - 1, item b from stream position first_item + b); else r [B][N] is read.
 *   f, fp both NULL: encrypt only.
 *   outputs, each optional (at least one): r_out [B][N] (the sampled r, to replay), e [B][N], value [B][N] (needs f, fp),
 *   packed [B][output_size][4] = packOutput(max, N, .) of the LAST stage's result: value with max = p - 1 when decrypting (then it
 *   comes out of the decrypt kernel itself: ntru_decrypt_pack_batch_dev), else e with max = q - 1 (out of the encrypt kernel itself when
 *   e is not asked for as well: ntru_encrypt_pack_batch_dev; sizes from ntru_pack_params). */
int ntru_pipeline_batch(ntru_engine_t *eng, int N, int q, int p, const uint16_t *h, const int8_t *f, const uint8_t *fp,
                        const uint32_t *key, uint64_t first_item, int n1, int n2, const uint8_t *r, const uint8_t *m, int64_t B,
                        uint8_t *r_out, uint16_t *e, uint8_t *value, uint64_t *packed);
/* Plain device buffers on the engine's device for use with the *_dev entry points.  ntru_dev_upload returns when `src` may be
 * reused and orders the data before every later call on the engine's stream; ntru_dev_download first waits for that stream. */
int ntru_dev_alloc(ntru_engine_t *eng, size_t bytes, void **d_ptr);
int ntru_dev_free(ntru_engine_t *eng, void *d_ptr);
int ntru_dev_upload(ntru_engine_t *eng, void *d_dst, const void *src, size_t bytes);
int ntru_dev_download(ntru_engine_t *eng, void *dst, const void *d_src, size_t bytes);

/* 1 if (N, mod) is a parameter set the kernels implement: mod a power of two <= 65536, or a small
 * modulus with N*(mod-1)^2 < 65536 (p = 3 for every NTRU set); 2 <= N <= NTRU_MAX_N. */
int ntru_engine_supports(int N, int mod);
#define NTRU_MAX_N 1920

/* ---- generic product + split: replaces multiplyPolynomials(a,b,mod) followed by
 *      dividePolynomials(.,I,mod) (index.js:319-355, 358-401) for per-item operands.
 *      a, b: [B][N] values (any uint16, reduced or not when mod is a power of two; < mod otherwise).
 *      quot[b][k] = (mod - c[N+k]) % mod,  rem[b][k] = (c[k] + c[N+k]) % mod,  c = linear product. */
int ntru_polymul_split(ntru_engine_t *eng, int N, int mod, const uint16_t *a, const uint16_t *b,
                       int64_t B, uint16_t *quot, uint16_t *rem);
int ntru_polymul_split_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b,
                           int64_t B, uint16_t *d_quot, uint16_t *d_rem);

/* ---- stand-alone split: dividePolynomials(a, I, mod) (index.js:358-401 with b = I = 1 - x^N) for dividends that
 *      are already reduced into [0,mod).  a: [B][2N] (row b = the dividend's coefficients 0..2N-1, zero padded).
 *      quot[b][k] = (mod - a[N+k]) % mod, rem[b][k] = (a[k] + a[N+k]) % mod   (SURVEY.md 0.3). */
int ntru_split_by_I(ntru_engine_t *eng, int N, int mod, const uint16_t *a, int64_t B, uint16_t *quot, uint16_t *rem);
int ntru_split_by_I_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, int64_t B, uint16_t *d_quot,
                        uint16_t *d_rem);

/* ---- coefficient-wise sum: addPolynomials(a, b, mod) (index.js:235-244) on [B][N] rows, e.g. the additive
 *      homomorphism on ciphertexts of test/reference.test.js:46-61.  out[b][k] = (a[b][k] + b[b][k]) % mod. */
int ntru_add_batch(ntru_engine_t *eng, int N, int mod, const uint16_t *a, const uint16_t *b, int64_t B, uint16_t *out);
int ntru_add_batch_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b, int64_t B,
                       uint16_t *d_out);

/* ---- on-device ternary sampler: generateCustomArray(N, n1, n2) (index.js:461-488) for B items, so that encryptBits'
 *      randomness r never crosses PCIe.  Row b (item index first_item + b) gets n1 ones, n2 entries equal to `other`
 *      (2 = p-1 for r after the index.js:89 map) and zeros, shuffled exactly like the reference: for i = N-1 .. 1:
 *      j = u32 % (i+1), swap -- with the u32 of step t taken from word t of the ChaCha20 keystream (RFC 8439 block
 *      function) under `key` (8 words) with nonce (item_lo, item_hi, 0x4e545255) and block counter 0,1,2,...
 *      A host replays the stream with any ChaCha20 implementation.  key is a HOST pointer in both variants. */
int ntru_sample_ternary(ntru_engine_t *eng, int N, int n1, int n2, int other, const uint32_t *key, uint64_t first_item,
                        int64_t B, uint8_t *out);
int ntru_sample_ternary_dev(ntru_engine_t *eng, int N, int n1, int n2, int other, const uint32_t *key,
                            uint64_t first_item, int64_t B, uint8_t *d_out);
/* Rounds of the sampler's block function: 20 (ChaCha20 as in RFC 8439: the default), 12 or 8 (ChaCha12 / ChaCha8: the same
 * quarter round, state and output rule, fewer double rounds).  generateCustomArray's contract (index.js:461-488) is the shuffle
 * order and its `u32 % (i + 1)` draws, not the generator behind crypto.getRandomValues; the sampler kernel is bound by the
 * vector issue of these rounds, and a host replays whichever variant it asked for.  Applies to ntru_sample_ternary[_dev] and to the
 * sampler stage of ntru_pipeline_batch of this engine until changed. */
int ntru_engine_set_sampler_rounds(ntru_engine_t *eng, int rounds);
int ntru_engine_get_sampler_rounds(ntru_engine_t *eng);

/* ---- encryptBits (index.js:87-110) for B plaintexts under one public key.
 *      h[N] in [0,q); r[B][N] in {0,1,2} (the sampler output with -1 already mapped to p-1, index.js:89);
 *      m[B][N] plaintext coefficients (README: 0/1/2; any byte is accepted and added mod q).
 *      e = remainderE (the ciphertext), quotE = quotientE (may be NULL when the witness is not wanted). */
int ntru_encrypt_batch(ntru_engine_t *eng, int N, int q, const uint16_t *h, const uint8_t *r, const uint8_t *m,
                       int64_t B, uint16_t *e, uint16_t *quotE);
int ntru_encrypt_batch_dev(ntru_engine_t *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r,
                           const uint8_t *d_m, int64_t B, uint16_t *d_e, uint16_t *d_quotE);

/* ---- decryptBits (index.js:111-140) for B ciphertexts under one private key.
 *      f[N] in {-1,0,1}; fp[N] in [0,p); e[B][N] in [0,q).
 *      value = remainder2; quot1 / rem1 / quot2 = quotient1 / remainder1 / quotient2 (each may be NULL).
 *      The centred lift is the reference's `x > q/2 ? (x+1)%p : x%p` verbatim. */
int ntru_decrypt_batch(ntru_engine_t *eng, int N, int q, int p, const int8_t *f, const uint8_t *fp,
                       const uint16_t *e, int64_t B, uint8_t *value, uint16_t *quot1, uint16_t *rem1,
                       uint8_t *quot2);
int ntru_decrypt_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f, const uint8_t *d_fp,
                           const uint16_t *d_e, int64_t B, uint8_t *d_value, uint16_t *d_quot1,
                           uint16_t *d_rem1, uint8_t *d_quot2);

/* The same two operations on PITCHED batch arrays: row b of every batch array (r, m, e, quotE / e, value, quot1, rem1,
 * quot2) starts at element b * ld of its array, ld >= N elements (so 2 * ld bytes for the uint16 arrays and ld bytes for
 * the byte arrays); each array holds B * ld elements, the ld - N pad elements of a row are neither read into the result
 * nor written.  ld == N is the dense layout of the entry points above.  A pitch that makes rows start on cache-line
 * boundaries (ld a multiple of 64) lets the result stores go out as whole lines: at N = 821 the store pattern alone
 * reaches 3.3 TB/s at ld = 832 against 2.4 TB/s at ld = 821; inside the kernels that is worth 5-7 % of encrypt and
 * 2-3 % of a round trip (EXPERIMENTS.md round 1, profiles/archive/r01_bench_row_pitch_ab.jsonl).  The reference has no memory
 * layout of its own (JS arrays, index.js:87-140), so the pitch is purely the host binding's choice.
 * Only the matrix-core kernels take a pitch: with ld != N the call fails with NTRU_ERR_UNSUPPORTED where they do not
 * apply (kernel path 1-3 forced, q > 8192, N or ld > 1024, p != 3). */
int ntru_encrypt_batch_pitched_dev(ntru_engine_t *eng, int N, int q, int ld, const uint16_t *d_h, const uint8_t *d_r,
                                   const uint8_t *d_m, int64_t B, uint16_t *d_e, uint16_t *d_quotE);
int ntru_decrypt_batch_pitched_dev(ntru_engine_t *eng, int N, int q, int p, int ld, const int8_t *d_f,
                                   const uint8_t *d_fp, const uint16_t *d_e, int64_t B, uint8_t *d_value,
                                   uint16_t *d_quot1, uint16_t *d_rem1, uint8_t *d_quot2);

/* loadPrivateKeyF / polyInv (index.js:30-49, 491-514) for B keys: fq[b] = f[b]^-1 in Z_q[x]/(x^N - 1) (q a power of two:
 * inverse modulo 2, then Newton rounds v <- 2v - f v^2), fp[b] = f[b]^-1 modulo p = 3; f in {-1,0,1}.  The inverse is
 * unique, so for units the result equals the reference's Euclidean algorithm bit for bit.  For f that is not a unit the
 * matching flag is set and the row is zero; the reference throws 'invalid_gcd' / 'invalid fq' for most such f but its
 * `&&` checks (index.js:41-45, :451) accept some and return meaningless polynomials -- that artefact is not reproduced.
 * Either of fq / fp may be NULL when only the other inverse is wanted (polyInv(f, I, 3) / polyInv(f, I, q)).
 * The Newton temporaries (4 x 2N bytes per key, at most 65536 keys at a time) live in an engine-owned buffer that only
 * grows; nothing is allocated per call and the _dev form does not synchronise.  [§8(f) #1] */
int ntru_invert_key_batch(ntru_engine_t *eng, int N, int q, int p, const int8_t *f, int64_t B, uint16_t *fq, uint8_t *fp,
                          uint8_t *flags);
int ntru_invert_key_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f, int64_t B, uint16_t *d_fq,
                              uint8_t *d_fp, uint8_t *d_flags);

/* generatePublicKeyH (index.js:72-79) for B keys: h[b] = remainder of ((p * fq[b]) mod q) * g[b] by 1 - x^N, mod q
 * (before trimPolynomial).  fq: mod-q inverse of f, g in {-1,0,1}; p*(q-1) must fit 16 bits.  Per-item operands on both
 * sides: one key per wavefront on the matrix cores (k_public_key_m) for 128 <= N <= 1024 and q <= 8192, the vector-ALU
 * product kernel otherwise.  [§8(f) #1, the part of key generation that is a product] */
int ntru_public_key_batch(ntru_engine_t *eng, int N, int q, int p, const uint16_t *fq, const int8_t *g, int64_t B,
                          uint16_t *h);
int ntru_public_key_batch_dev(ntru_engine_t *eng, int N, int q, int p, const uint16_t *d_fq, const int8_t *d_g,
                              int64_t B, uint16_t *d_h);

/* ---- verifyKeysInputs (index.js:141-197) for B independent key pairs (per-item operands).
 *      f, g [B][N] in {-1,0,1}; fq, h [B][N] in [0,q); fp [B][N] in [0,p).
 *      Three witnesses per item: fq*f mod q, fp*f mod p, (p*fq)*g mod q, each as quotient + remainder;
 *      flags[B] gets the NTRU_FLAG_* bits.  One key pair per wavefront; on the matrix cores where the note at the top
 *      of this file says so (k_verify_keys_m), otherwise on the ternary add path / the multiply-accumulate kernels. */
int ntru_verify_keys_batch(ntru_engine_t *eng, int N, int q, int p, const int8_t *f, const int8_t *g,
                           const uint16_t *fq, const uint8_t *fp, const uint16_t *h, int64_t B,
                           uint16_t *quot_fq, uint16_t *rem_fq, uint8_t *quot_fp, uint8_t *rem_fp,
                           uint16_t *quot_h, uint16_t *rem_h, uint8_t *flags);
int ntru_verify_keys_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f, const int8_t *d_g,
                               const uint16_t *d_fq, const uint8_t *d_fp, const uint16_t *d_h, int64_t B,
                               uint16_t *d_quot_fq, uint16_t *d_rem_fq, uint8_t *d_quot_fp, uint8_t *d_rem_fp,
                               uint16_t *d_quot_h, uint16_t *d_rem_h, uint8_t *d_flags);

/* ---- pinned host memory.  Buffers from ntru_host_alloc are page-locked: the host-pointer entry points below DMA
 *      straight from / to them (any other host memory is staged through the engine's own pinned arenas, which costs a
 *      CPU copy per byte).  The N-API addon hands them to JavaScript as TypedArrays (allocUint8 / allocUint16). */
void *ntru_host_alloc(size_t bytes);
void ntru_host_free(void *p);

/* ---- several devices in one process (contiguous batch shards, no exchange between devices).  One engine
 *      per listed device id (an id may be listed more than once: every entry is its own engine with its own streams), one
 *      host thread per engine for the duration of a call; device g gets the items [g B / G, (g + 1) B / G).  Host pointers,
 *      same meaning as the single-device entry points above; on failure the message names the shard.  (bench.py's multi-GPU
 *      mode is one process per GPU under torchrun instead; this is for a host that owns all GPUs of a node, e.g. Node.js.) */
typedef struct ntru_multi ntru_multi_t;
int ntru_multi_create(const int *device_ids, int n_dev, ntru_multi_t **out);
void ntru_multi_destroy(ntru_multi_t *m);
int ntru_multi_engines(const ntru_multi_t *m);
int ntru_multi_encrypt_batch(ntru_multi_t *m, int N, int q, const uint16_t *h, const uint8_t *r, const uint8_t *mm, int64_t B,
                             uint16_t *e, uint16_t *quotE);
int ntru_multi_decrypt_batch(ntru_multi_t *m, int N, int q, int p, const int8_t *f, const uint8_t *fp, const uint16_t *e,
                             int64_t B, uint8_t *value, uint16_t *quot1, uint16_t *rem1, uint8_t *quot2);
int ntru_multi_verify_keys_batch(ntru_multi_t *m, int N, int q, int p, const int8_t *f, const int8_t *g, const uint16_t *fq,
                                 const uint8_t *fp, const uint16_t *h, int64_t B, uint16_t *quot_fq, uint16_t *rem_fq,
                                 uint8_t *quot_fp, uint8_t *rem_fp, uint16_t *quot_h, uint16_t *rem_h, uint8_t *flags);
int ntru_multi_polymul_split(ntru_multi_t *m, int N, int mod, const uint16_t *a, const uint16_t *b, int64_t B, uint16_t *quot,
                             uint16_t *rem);
int ntru_multi_invert_key_batch(ntru_multi_t *m, int N, int q, int p, const int8_t *f, int64_t B, uint16_t *fq, uint8_t *fp,
                                uint8_t *flags);
int ntru_multi_public_key_batch(ntru_multi_t *m, int N, int q, int p, const uint16_t *fq, const int8_t *g, int64_t B,
                                uint16_t *h);

/* ---- generic, reference-faithful family (ntru_generic.hip): the reference's own algorithms on int64 coefficients, for
 *      what the fast kernels do not cover -- moduli above 65536 (multiplyPolynomials(a, b, 2^20), test/circuits.test.js:72),
 *      arbitrary divisors (dividePolynomials, index.js:358-401, e.g. test/circuits.test.js:165-170), and the stand-alone
 *      extendedEuclideanAlgorithm (index.js:425-459) / polyInv (index.js:491-514) incl. their behaviour on non-units.
 *      a: [B][la], b: [B][lb] signed, possibly unreduced coefficients, |x| <= 2^26; 1 <= mod <= 2^26 (every product the
 *      reference forms then stays exact in a JS double).  Every item of a batch has the same operand lengths la, lb --
 *      lengths matter to the reference (a.length, `r0.length !== 1`), so callers must not pad.  Result rows have a pitch of
 *      ntru_generic_capacity(la, lb) elements; *_len[b] is the length of the (trimmed) result array the reference returns.
 *      status[b]: 0, or the NTRU_GENERIC_* code of the error the reference throws for item b (its results are then empty).
 *      Host pointers; one item per wavefront; a latency path, not a throughput path. */
#define NTRU_GENERIC_DIV_BY_ZERO 1  /* "Cannot divide by zero polynomial."  (index.js:360) */
#define NTRU_GENERIC_NO_INVERSE 2   /* "No inverse exists for division."    (index.js:378) */
#define NTRU_GENERIC_INVALID_GCD 3  /* "invalid_gcd"                        (index.js:452) */
#define NTRU_GENERIC_CAPACITY 4     /* internal: a result outgrew the work area (never expected) */
int ntru_generic_capacity(int la, int lb);
/* multiplyPolynomials(a, b, mod): out[b] = the linear product, coefficients in [0, mod), trimmed ([0] when la or lb is 0). */
int ntru_generic_multiply(ntru_engine_t *eng, int la, int lb, int64_t mod, const int64_t *a, const int64_t *b, int64_t B,
                          int64_t *out, int32_t *out_len);
/* dividePolynomials(a, b, mod) -> { quotient, remainder } exactly as index.js:358-401 computes them. */
int ntru_generic_divide(ntru_engine_t *eng, int la, int lb, int64_t mod, const int64_t *a, const int64_t *b, int64_t B,
                        int64_t *quot, int32_t *quot_len, int64_t *rem, int32_t *rem_len, uint8_t *status);
/* extendedEuclideanAlgorithm(a, b, mod) -> { gcd, inverse } (index.js:425-459). */
int ntru_generic_eea(ntru_engine_t *eng, int la, int lb, int64_t mod, const int64_t *a, const int64_t *b, int64_t B,
                     int64_t *gcd, int32_t *gcd_len, int64_t *inverse, int32_t *inverse_len, uint8_t *status);
/* polyInv(a, polyI, mod) (index.js:491-514): EEA modulo 2 + log2(mod) - 1 Newton rounds when mod is a power of two,
 * plain EEA otherwise. */
int ntru_generic_poly_inv(ntru_engine_t *eng, int la, int lb, int64_t mod, const int64_t *a, const int64_t *poly_i,
                          int64_t B, int64_t *inverse, int32_t *inverse_len, uint8_t *status);

/* ---- BN254 field-element packing: packOutput / unpackInput (index.js:572-620), the wire format of the circuits'
 *      CombineArray / UnpackArray (circuits/ntru.circom:258-306).  bits = floor(log2(max_val)+1) per value,
 *      per_output = floor(252/bits) values per field element, arr_len / output_size as index.js:575-580.
 *      A field element is four little-endian uint64 limbs.  1 <= max_val <= 65535.
 *      pack:   data [B][data_len] -> out [B][output_size][4];   out[b][o] = sum_j data[b][o*per+j] << (j*bits)
 *      unpack: in [B][packed_size][4] -> out [B][packed_size*per], per = floor(packed_bits/bits) (trimming is host glue) */
int ntru_pack_params(int max_val, int data_len, int *bits, int *per_output, int *arr_len, int *output_size);
int ntru_pack_batch(ntru_engine_t *eng, int max_val, int data_len, const uint16_t *data, int64_t B, uint64_t *out);
int ntru_pack_batch_dev(ntru_engine_t *eng, int max_val, int data_len, const uint16_t *d_data, int64_t B, uint64_t *d_out);
/* packOutput of an array of BYTES (values <= 255: decryptBits' value and quotient2, r, m) without widening it to uint16 first. */
int ntru_pack_bytes_batch_dev(ntru_engine_t *eng, int max_val, int data_len, const uint8_t *d_data, int64_t B, uint64_t *d_out);
/* decryptBits (index.js:111-140) + packOutput(p - 1, N, value) (index.js:572-596) of its result, value-only mode, device pointers:
 * d_packed [B][output_size][4] (sizes from ntru_pack_params(p - 1, N, ...)).  Where the matrix-core decrypt applies (shared key, p == 3,
 * q <= 8192, N <= 1024, d_packed 16-byte aligned) AND the kernel's 2-bit image of the values fits behind its mod-p tables in the LDS
 * (it does from N of about 100 up; every BASELINE size) this is ONE kernel -- the field elements come straight out of the second
 * product's epilogue and d_value may be NULL (nothing but the packed rows is written); elsewhere it is decrypt + pack, d_value is
 * needed as the intermediate and a NULL d_value is NTRU_ERR_ARG. */
int ntru_decrypt_pack_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f, const uint8_t *d_fp,
                                const uint16_t *d_e, int64_t B, uint8_t *d_value, uint64_t *d_packed);
/* encryptBits (index.js:87-110) + packOutput(q - 1, N, e) (index.js:572-596) of its result, device pointers: d_packed
 * [B][output_size][4] (sizes from ntru_pack_params(q - 1, N, ...)).  With d_e == NULL and where the row-image matrix kernel applies
 * (shared key, q in {2048, 4096, 8192}, N <= 1024, d_packed 16-byte aligned) this is ONE kernel that writes nothing but the packed
 * rows (32 output_size bytes per ciphertext instead of 2 N + 32 output_size); with d_e != NULL, or elsewhere, it is encrypt (e only)
 * + pack, and d_e is needed. */
int ntru_encrypt_pack_batch_dev(ntru_engine_t *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m,
                                int64_t B, uint16_t *d_e, uint64_t *d_packed);
int ntru_unpack_batch(ntru_engine_t *eng, int max_val, int packed_bits, const uint64_t *in, int packed_size, int64_t B,
                      uint16_t *out);
int ntru_unpack_batch_dev(ntru_engine_t *eng, int max_val, int packed_bits, const uint64_t *d_in, int packed_size,
                          int64_t B, uint16_t *d_out);

#ifdef __cplusplus
}
#endif
#endif /* NTRU_ENGINE_H */
