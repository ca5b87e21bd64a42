// abi.hip -- engine life cycle and the *_dev entry points of include/ntru_engine.h that choose between kernel families.  The kernels
// and their launchers live in valu_families.hip, matrix_encrypt.hip, matrix_decrypt.hip, matrix_peritem.hip; the entry points of key
// inversion, sampler, packing and the elementwise kernels are in keygen_sampler_pack.hip, the host-pointer forms in ntru_host.hip.
//
// No CPU fallback exists in this library: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "engine_internal.h"

static thread_local std::string g_err;
int ntru_fail(int code, const std::string &msg) { g_err = msg; return code; }

int ntru_grow_dev(GrowBuf *b, size_t bytes) {
  if (bytes <= b->cap) return NTRU_OK;
  if (b->p) (void)hipFree(b->p);          // waits for the device: nothing in flight still reads the old buffer
  b->p = nullptr; b->cap = 0;
  const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  if (hipMalloc(&b->p, want) != hipSuccess) { b->p = nullptr; return fail(NTRU_ERR_HIP, "hipMalloc failed"); }
  b->cap = want;
  return NTRU_OK;
}

int ntru_grow_pinned(GrowBuf *b, size_t bytes) {
  if (bytes <= b->cap) return NTRU_OK;
  if (b->p) (void)hipHostFree(b->p);
  b->p = nullptr; b->cap = 0;
  const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  if (hipHostMalloc(&b->p, want, hipHostMallocDefault) != hipSuccess) { b->p = nullptr; return fail(NTRU_ERR_HIP, "hipHostMalloc failed"); }
  b->cap = want;
  return NTRU_OK;
}

// Occupancy of (kernel, block size, LDS bytes), asked once per engine; the first use of a kernel with more than 64 KiB
// of dynamic LDS also raises its limit.
int ntru_blocks_per_cu(ntru_engine *eng, const void *fn, int threads, size_t lds, int *per_cu) {
  for (int i = 0; i < eng->n_occ; i++)
    if (eng->occ[i].fn == fn && eng->occ[i].lds == lds && eng->occ[i].threads == threads) { *per_cu = eng->occ[i].per_cu; return NTRU_OK; }
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int n = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds));
  // The query over-reports LDS-bound residency: a CU hands out its 160 KB in pieces of 1280 bytes per WORKGROUP (bench_micro/wg_residency:
  // twelve workgroups of 13312 bytes are reported, eleven are resident; twenty of 8192, eighteen).  A persistent grid sized by the
  // reported figure runs its surplus workgroups as a second round on a nearly empty chip.
  if (lds > 0 && getenv("NTRU_TRUST_OCCUPANCY_QUERY") == nullptr) {
    const size_t piece = 1280, per_wg = (lds + piece - 1) / piece * piece;
    const int fit = (int)((size_t)160 * 1024 / per_wg);
    if (fit < n) n = fit;
  }
  if (getenv("NTRU_DEBUG_OCC")) fprintf(stderr, "occupancy: fn=%p threads=%d lds=%zu -> %d workgroups per CU\n", fn, threads, lds, n);
  if (n < 1) n = 1;
  if (eng->n_occ < (int)(sizeof eng->occ / sizeof eng->occ[0])) eng->occ[eng->n_occ++] = {fn, lds, threads, n};
  *per_cu = n;
  return NTRU_OK;
}

// The engine-owned scratch buffer is shared by every *_dev call that needs temporaries.  Calls on ONE stream are ordered by the
// stream; a call on another stream first waits (on the device, not the host) for the event recorded behind the last user.
int ntru_scratch_acquire(ntru_engine *eng, size_t bytes, char **p) {
  GrowBuf *b = eng->cur_scratch;
  if (b == &eng->scratch_dev && eng->scratch_used && eng->scratch_stream != eng->stream)
    HIP_TRY(hipStreamWaitEvent(eng->stream, eng->scratch_event, 0));
  if (int rc = ntru_grow_dev(b, bytes)) return rc;
  *p = (char *)b->p;
  return NTRU_OK;
}

int ntru_scratch_release(ntru_engine *eng) {
  if (eng->cur_scratch != &eng->scratch_dev) return NTRU_OK;      // a host-path slot: its own stream, its own buffer
  if (!eng->scratch_event) HIP_TRY(hipEventCreateWithFlags(&eng->scratch_event, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(eng->scratch_event, eng->stream));
  eng->scratch_stream = eng->stream;
  eng->scratch_used = true;
  return NTRU_OK;
}

extern "C" int ntru_engine_supports(int N, int mod) {
  if (N < 2 || N > NTRU_MAX_N || mod < 2) return 0;
  if (is_pow2(mod)) return mod <= 65536;
  return (long)N * (mod - 1) * (mod - 1) < 65536;
}

extern "C" int ntru_engine_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char *ntru_last_error(void) { return g_err.c_str(); }

extern "C" int ntru_engine_create(int device, ntru_engine_t **out) {
  if (!out) return fail(NTRU_ERR_ARG, "ntru_engine_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(NTRU_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) +
                                        "); this engine has no CPU fallback");
  if (device < 0 || device >= n) return fail(NTRU_ERR_NO_DEVICE, "device id out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  ntru_engine *eng = new ntru_engine();
  eng->device = device;
  eng->stream = nullptr;
  eng->cus = prop.multiProcessorCount;
  eng->path = 0;
  eng->sampler_rounds = 20;
  eng->last_kernel[0] = 0;
  eng->n_occ = 0;
  eng->cur_scratch = &eng->scratch_dev;
  eng->st_up = eng->st_comp = eng->st_down = nullptr;
  eng->st_aux = nullptr;
  eng->ev_fork = eng->ev_join = nullptr;
  eng->scratch_stream = nullptr;
  eng->scratch_event = nullptr;
  eng->scratch_used = false;
  eng->max_blocks_per_cu = 0;
  if (const char *cap = getenv("NTRU_MAX_BLOCKS_PER_CU")) {      // tuning experiments only; read once
    const int c = atoi(cap);
    if (c >= 1) eng->max_blocks_per_cu = c;
  }
  *out = eng;
  return NTRU_OK;
}

extern "C" void ntru_engine_destroy(ntru_engine_t *eng) {
  if (!eng) return;
  (void)hipSetDevice(eng->device);
  for (hipStream_t st : {eng->st_up, eng->st_comp, eng->st_down, eng->st_aux})
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (hipEvent_t ev : {eng->ev_fork, eng->ev_join}) if (ev) (void)hipEventDestroy(ev);
  for (HostSlot &s : eng->slot) {
    for (hipEvent_t ev : {s.up_done, s.comp_done, s.down_done}) if (ev) (void)hipEventDestroy(ev);
    if (s.pinned.p) (void)hipHostFree(s.pinned.p);
    if (s.dev.p) (void)hipFree(s.dev.p);
    if (s.scratch.p) (void)hipFree(s.scratch.p);
  }
  if (eng->shared_dev.p) (void)hipFree(eng->shared_dev.p);
  if (eng->scratch_dev.p) (void)hipFree(eng->scratch_dev.p);
  if (eng->scratch_event) (void)hipEventDestroy(eng->scratch_event);
  delete eng;
}

extern "C" int ntru_engine_set_stream(ntru_engine_t *eng, void *hip_stream) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  eng->stream = (hipStream_t)hip_stream;
  return NTRU_OK;
}

#ifdef NTRU_EXPERIMENTS
static const int kMaxKernelPath = 12;
#else
static const int kMaxKernelPath = 5;
#endif

extern "C" int ntru_engine_set_kernel_path(ntru_engine_t *eng, int path) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (path < 0 || path > kMaxKernelPath)
    return fail(NTRU_ERR_ARG, "kernel path must be 0 (auto), 1 (MAC), 2 (add), 3 (add without dot8), 4 (matrix cores, two workgroups per CU) or "
                              "5 (matrix cores, lock-step decrypt); 6 (role-split encrypt), 7 (chunked encrypt stores), 8 (direct-to-LDS decrypt), "
                              "9 (lock-step encrypt), 10 (row-image encrypt), 11 (decrypt with an fp4 second product) and 12 (verify_keys on the 16-row matrix tile) exist only in a library built with -DNTRU_EXPERIMENTS (make experiments)");
  eng->path = path;
  return NTRU_OK;
}

extern "C" int ntru_engine_set_sampler_rounds(ntru_engine_t *eng, int rounds) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (rounds != 20 && rounds != 12 && rounds != 8)
    return fail(NTRU_ERR_ARG, "sampler rounds must be 20 (ChaCha20, RFC 8439: the default), 12 or 8");
  eng->sampler_rounds = rounds;
  return NTRU_OK;
}

extern "C" int ntru_engine_get_sampler_rounds(ntru_engine_t *eng) { return eng ? eng->sampler_rounds : 0; }

extern "C" const char *ntru_engine_last_kernel(ntru_engine_t *eng) { return eng ? eng->last_kernel : ""; }

extern "C" int ntru_engine_synchronize(ntru_engine_t *eng) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  HIP_TRY(hipSetDevice(eng->device));
  HIP_TRY(hipStreamSynchronize(eng->stream));
  return NTRU_OK;
}

int ntru_check_common(const ntru_engine *eng, int N, int q, long B) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0) return fail(NTRU_ERR_ARG, "negative batch size");
  if (!is_pow2(q) || !ntru_engine_supports(N, q))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported (N, q): need 2 <= N <= 1920 and q a power of two <= 65536");
  return NTRU_OK;
}

static int check_pitch(int N, int ld) {
  if (ld < N || ld > 1024) return fail(NTRU_ERR_ARG, "row pitch must satisfy N <= ld <= 1024 elements");
  return NTRU_OK;
}
static const char *const kPitchedOnly = "a row pitch other than N needs the matrix-core kernels (kernel path 0 or 4, q <= 8192, N <= 1024, p == 3)";

extern "C" int ntru_encrypt_batch_dev(ntru_engine_t *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r,
                                      const uint8_t *d_m, int64_t B, uint16_t *d_e, uint16_t *d_quotE) {
  return ntru_encrypt_batch_pitched_dev(eng, N, q, N, d_h, d_r, d_m, B, d_e, d_quotE);
}

extern "C" int ntru_encrypt_batch_pitched_dev(ntru_engine_t *eng, int N, int q, int ld, const uint16_t *d_h,
                                              const uint8_t *d_r, const uint8_t *d_m, int64_t B, uint16_t *d_e,
                                              uint16_t *d_quotE) {
  if (int rc = ntru_check_common(eng, N, q, B)) return rc;
  if (ld != N) if (int rc = check_pitch(N, ld)) return rc;
  if (B == 0) return NTRU_OK;
  if (!d_h || !d_r || !d_m || !d_e) return fail(NTRU_ERR_ARG, "ntru_encrypt_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  if (eng->path == 10) {                                       // row-image kernel (matrix_rowimage.hip) where it applies
    const int rcw = ntru_launch_encrypt_rowimage(eng, N, q, ld, d_h, d_r, d_m, B, d_e, d_quotE);
    if (rcw != NTRU_NOT_TAKEN) return rcw;
  }
  const int rc = ntru_launch_encrypt_matrix(eng, N, q, ld, d_h, d_r, d_m, B, d_e, d_quotE);      // shared key: batch x Toeplitz on the matrix cores
  if (rc != NTRU_NOT_TAKEN) return rc;
  if (ld != N) return fail(NTRU_ERR_UNSUPPORTED, kPitchedOnly);
  return ntru_launch_encrypt_valu(eng, N, q, d_h, d_r, d_m, B, d_e, d_quotE);
}

extern "C" int ntru_decrypt_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f,
                                      const uint8_t *d_fp, const uint16_t *d_e, int64_t B, uint8_t *d_value,
                                      uint16_t *d_quot1, uint16_t *d_rem1, uint8_t *d_quot2) {
  return ntru_decrypt_batch_pitched_dev(eng, N, q, p, N, d_f, d_fp, d_e, B, d_value, d_quot1, d_rem1, d_quot2);
}

extern "C" int ntru_decrypt_batch_pitched_dev(ntru_engine_t *eng, int N, int q, int p, int ld, const int8_t *d_f,
                                              const uint8_t *d_fp, const uint16_t *d_e, int64_t B, uint8_t *d_value,
                                              uint16_t *d_quot1, uint16_t *d_rem1, uint8_t *d_quot2) {
  if (int rc = ntru_check_common(eng, N, q, B)) return rc;
  if (ld != N) if (int rc = check_pitch(N, ld)) return rc;
  if (is_pow2(p) || !ntru_engine_supports(N, p))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported p: need a small non-power-of-two modulus with N*(p-1)^2 < 65536");
  if (B == 0) return NTRU_OK;
  if (!d_f || !d_fp || !d_e || !d_value) return fail(NTRU_ERR_ARG, "ntru_decrypt_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  const int rc = ntru_launch_decrypt_matrix(eng, N, q, p, ld, d_f, d_fp, d_e, B, d_value, d_quot1, d_rem1, d_quot2);
  if (rc != NTRU_NOT_TAKEN) return rc;
  if (ld != N) return fail(NTRU_ERR_UNSUPPORTED, kPitchedOnly);
  return ntru_launch_decrypt_valu(eng, N, q, p, d_f, d_fp, d_e, B, d_value, d_quot1, d_rem1, d_quot2);
}

// decryptBits (index.js:111-140) followed by packOutput(p - 1, N, value) (index.js:572-596), value-only mode: ONE kernel where the
// matrix path applies (the packed field elements come out of the second product's epilogue), else decrypt + pack as two launches
// (then d_value is needed as the intermediate: NTRU_ERR_ARG without it).
extern "C" int ntru_decrypt_pack_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f, const uint8_t *d_fp,
                                           const uint16_t *d_e, int64_t B, uint8_t *d_value, uint64_t *d_packed) {
  if (int rc = ntru_check_common(eng, N, q, B)) return rc;
  if (is_pow2(p) || !ntru_engine_supports(N, p))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported p: need a small non-power-of-two modulus with N*(p-1)^2 < 65536");
  int bits, per, al, os;
  if (int rc = ntru_pack_params(p - 1, N, &bits, &per, &al, &os)) return rc;
  if (B == 0) return NTRU_OK;
  if (!d_f || !d_fp || !d_e || !d_packed) return fail(NTRU_ERR_ARG, "ntru_decrypt_pack_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  if (eng->path == 0 || eng->path >= 4) {
    const int rc = ntru_launch_decrypt_pack_matrix(eng, N, q, p, d_f, d_fp, d_e, B, d_value, d_packed, os);
    if (rc != NTRU_NOT_TAKEN) return rc;
  }
  if (!d_value) return fail(NTRU_ERR_ARG, "ntru_decrypt_pack_batch: outside the fused kernel's range d_value is needed as the intermediate");
  if (int rc = ntru_decrypt_batch_dev(eng, N, q, p, d_f, d_fp, d_e, B, d_value, nullptr, nullptr, nullptr)) return rc;
  return ntru_pack_bytes_batch_dev(eng, p - 1, N, d_value, B, d_packed);
}

// encryptBits (index.js:87-110) followed by packOutput(q - 1, N, e) (index.js:572-596): ONE kernel where the row-image matrix kernel
// applies and the caller does not ask for e itself (d_e == NULL: nothing but the packed rows is written), else encrypt + pack as two
// launches (then d_e is the intermediate: NTRU_ERR_ARG without it).
extern "C" int ntru_encrypt_pack_batch_dev(ntru_engine_t *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m,
                                           int64_t B, uint16_t *d_e, uint64_t *d_packed) {
  if (int rc = ntru_check_common(eng, N, q, B)) return rc;
  int bits, per, al, os;
  if (int rc = ntru_pack_params(q - 1, N, &bits, &per, &al, &os)) return rc;
  if (B == 0) return NTRU_OK;
  if (!d_h || !d_r || !d_m || !d_packed) return fail(NTRU_ERR_ARG, "ntru_encrypt_pack_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  if (!d_e && (eng->path == 0 || eng->path >= 4)) {
    const int rc = ntru_launch_encrypt_pack_rowimage(eng, N, q, d_h, d_r, d_m, B, d_packed, os);
    if (rc != NTRU_NOT_TAKEN) return rc;
  }
  if (!d_e) return fail(NTRU_ERR_ARG, "ntru_encrypt_pack_batch: outside the fused kernel's range d_e is needed as the intermediate");
  if (int rc = ntru_encrypt_batch_dev(eng, N, q, d_h, d_r, d_m, B, d_e, nullptr)) return rc;
  return ntru_pack_batch_dev(eng, q - 1, N, d_e, B, d_packed);
}

extern "C" int ntru_polymul_split_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b,
                                      int64_t B, uint16_t *d_quot, uint16_t *d_rem) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0) return fail(NTRU_ERR_ARG, "negative batch size");
  if (!ntru_engine_supports(N, mod))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported (N, mod): mod must be a power of two <= 65536 or satisfy N*(mod-1)^2 < 65536");
  if (B == 0) return NTRU_OK;
  if (!d_a || !d_b || !d_quot || !d_rem) return fail(NTRU_ERR_ARG, "ntru_polymul_split: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  const int rc = ntru_launch_polymul_matrix(eng, N, mod, d_a, d_b, B, d_quot, d_rem);      // per-item product on the matrix cores
  if (rc != NTRU_NOT_TAKEN) return rc;
  return ntru_launch_polymul_valu(eng, N, mod, d_a, d_b, B, d_quot, d_rem);
}

extern "C" int ntru_public_key_batch_dev(ntru_engine_t *eng, int N, int q, int p, const uint16_t *d_fq,
                                         const int8_t *d_g, int64_t B, uint16_t *d_h) {
  if (int rc = ntru_check_common(eng, N, q, B)) return rc;
  if (p < 1 || (long)p * (q - 1) >= 65536) return fail(NTRU_ERR_UNSUPPORTED, "p*(q-1) must fit 16 bits");
  if (B == 0) return NTRU_OK;
  if (!d_fq || !d_g || !d_h) return fail(NTRU_ERR_ARG, "ntru_public_key_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  if (ntru_product_tern_matrix_applies(eng, N, q)) {
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_public_key_m");
    return ntru_launch_product_tern_matrix(eng, N, q, (uint32_t)p, d_fq, d_g, (long)B, d_h);
  }
  return ntru_launch_public_key_valu(eng, N, q, p, d_fq, d_g, B, d_h);
}

extern "C" int ntru_verify_keys_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f,
                                          const int8_t *d_g, const uint16_t *d_fq, const uint8_t *d_fp,
                                          const uint16_t *d_h, int64_t B, uint16_t *d_quot_fq, uint16_t *d_rem_fq,
                                          uint8_t *d_quot_fp, uint8_t *d_rem_fp, uint16_t *d_quot_h,
                                          uint16_t *d_rem_h, uint8_t *d_flags) {
  if (int rc = ntru_check_common(eng, N, q, B)) return rc;
  if (is_pow2(p) || !ntru_engine_supports(N, p))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported p: need a small non-power-of-two modulus with N*(p-1)^2 < 65536");
  if ((long)(q - 1) * p > 65535) return fail(NTRU_ERR_UNSUPPORTED, "p*(q-1) must fit 16 bits");
  if (B == 0) return NTRU_OK;
  if (!d_f || !d_g || !d_fq || !d_fp || !d_h || !d_quot_fq || !d_rem_fq || !d_quot_fp || !d_rem_fp || !d_quot_h ||
      !d_rem_h || !d_flags)
    return fail(NTRU_ERR_ARG, "ntru_verify_keys_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  // matrix-core kernel for per-item keys: p == 3, q <= 8192 (two int8 digit planes), 64 <= N <= 1024; automatic from N = 128
  const int rc = ntru_launch_verify_keys_matrix(eng, N, q, p, d_f, d_g, d_fq, d_fp, d_h, B, d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp,
                                                d_quot_h, d_rem_h, d_flags);
  if (rc != NTRU_NOT_TAKEN) return rc;
  return ntru_launch_verify_keys_valu(eng, N, q, p, d_f, d_g, d_fq, d_fp, d_h, B, d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp, d_quot_h,
                                      d_rem_h, d_flags);
}
