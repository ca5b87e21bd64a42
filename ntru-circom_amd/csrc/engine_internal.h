// engine_internal.h -- what the translation units of libntru_engine.so share (not part of the C ABI).
//
//   abi.hip                  engine life cycle + the *_dev entry points that choose between kernel families
//   valu_families.hip        vector-ALU families 1-3 (one ciphertext per wavefront): MAC, ternary add path, shared stepping + dot8
//   matrix_encrypt.hip       family 4, encryptBits with a shared key on the int8 matrix cores (k_encrypt_m, k_encrypt_md)
//   matrix_decrypt.hip       family 4, decryptBits with a shared key (k_decrypt_m, k_decrypt_m8)
//   matrix_peritem.hip       family 4 with per-item operands (k_verify_keys_m, k_polymul_m, k_product_tern_m)
//   keygen_sampler_pack.hip  key inversion, ternary sampler, field packing, elementwise kernels + their *_dev entry points
//   ntru_host.hip            host-pointer entry points: pinned staging, three stage streams, chunked H2D / kernel / D2H pipeline
//   ntru_generic.hip         reference-faithful generic family (arbitrary divisors, moduli up to 2^26, signed coefficients)
// Every kernel family exports the host function that launches it (ntru_launch_*, hidden visibility); a launcher returns
// NTRU_NOT_TAKEN when the parameters are outside its family's range and the dispatcher in abi.hip tries the next one.
#ifndef NTRU_ENGINE_INTERNAL_H
#define NTRU_ENGINE_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "ntru_engine.h"

#define NTRU_HIDDEN __attribute__((visibility("hidden")))

// Per-thread message behind ntru_last_error(); returns `code`.
NTRU_HIDDEN int ntru_fail(int code, const std::string &msg);

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return ntru_fail(NTRU_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
  } while (0)

// A buffer that only grows: allocated on first use, reused by every later call, released with the engine.
struct GrowBuf {
  void *p = nullptr;
  size_t cap = 0;
};

// Host-path staging (ntru_host.hip): a chunk moves through three STAGES, each with its own engine-owned stream -- upload (H2D),
// compute (the *_dev launches), download (D2H) -- chained by events, so that at any time ONE upload, one set of kernels and ONE
// download are in flight: chunk k+1 goes up while chunk k computes and chunk k-1 comes down (PCIe is full duplex: 57 GB/s one
// way, 47 + 47 GB/s both ways on this box, profiles/archive/r03_pcie_duplex.json).  Chunk k owns buffer set k % 3 (pinned host arena +
// device arena + scratch) until its download has finished.  [Round 2 ran each chunk on the stream of one of TWO slots: the two
// slots fell into lock-step -- both uploading, then both computing, then both downloading -- and the two directions never
// overlapped: 56 GB/s in total.]
struct HostSlot {
  GrowBuf pinned;           // hipHostMalloc
  GrowBuf dev;              // hipMalloc
  GrowBuf scratch;          // temporaries of a *_dev call of the chunk that owns this set
  hipEvent_t up_done = nullptr, comp_done = nullptr, down_done = nullptr;
  bool busy = false;        // a chunk has been enqueued with this set and its download has not been waited for yet
};
constexpr int NTRU_HOST_SLOTS = 3;

struct ntru_engine {
  int device;
  hipStream_t stream;       // caller's stream for the *_dev entry points (never owned)
  int cus;
  int path;                 // ntru_engine_set_kernel_path: 0 auto, 1 MAC kernels, 2 add path, 3 add path without dot8, 4 matrix cores as two
                            // workgroups per CU, 5 matrix cores with the lock-step decrypt; 6-9 only in -DNTRU_EXPERIMENTS builds
  int max_blocks_per_cu;    // NTRU_MAX_BLOCKS_PER_CU read once at creation (tuning experiments); 0 = no cap
  int sampler_rounds;       // ntru_engine_set_sampler_rounds: 20 (RFC 8439, default), 12 or 8 rounds of the sampler's ChaCha block function
  char last_kernel[64];     // name of the kernel the last *_dev call launched (reporting only)
  HostSlot slot[NTRU_HOST_SLOTS];
  hipStream_t st_up, st_comp, st_down;   // the three stage streams of the host path (created at first use)
  hipStream_t st_aux;       // forked from / joined back into the caller's stream inside ONE *_dev call (key inversion: the mod-p inversion,
  hipEvent_t ev_fork, ev_join;   // vector-ALU bound, runs beside the Newton chain on the matrix cores); created at first use
  GrowBuf shared_dev;       // shared key rows of the host path (h, f, fp)
  GrowBuf scratch_dev;      // temporaries of *_dev calls (Newton rounds, generic family) on the caller's stream
  hipStream_t scratch_stream;   // the stream whose work used scratch_dev last, and an event recorded behind that work: a call on
  hipEvent_t scratch_event;     // ANOTHER stream waits for it before it touches the buffer (ntru_scratch_acquire / _release)
  bool scratch_used;
  GrowBuf *cur_scratch;     // = &scratch_dev, or a slot's scratch while the host path borrows the engine for that slot
  // (kernel function, LDS bytes, block size) -> co-resident blocks per CU, filled at first use
  struct OccEntry { const void *fn; size_t lds; int threads; int per_cu; };
  OccEntry occ[64];
  int n_occ;
};

// Grows `b` to at least `bytes` of device memory (contents are not preserved).  Growing frees the old buffer, and hipFree waits
// for the whole device.
NTRU_HIDDEN int ntru_grow_dev(GrowBuf *b, size_t bytes);
// Grows `b` to at least `bytes` of pinned host memory.
NTRU_HIDDEN int ntru_grow_pinned(GrowBuf *b, size_t bytes);
// hipOccupancyMaxActiveBlocksPerMultiprocessor, cached per engine.
NTRU_HIDDEN int ntru_blocks_per_cu(ntru_engine *eng, const void *fn, int threads, size_t lds, int *per_cu);
// eng->cur_scratch grown to `bytes` for work about to be enqueued on eng->stream.  If the previous user of the engine-owned
// buffer ran on another stream, eng->stream first waits for that work (two *_dev calls on different streams no longer race).
NTRU_HIDDEN int ntru_scratch_acquire(ntru_engine *eng, size_t bytes, char **p);
// Marks the end of the work enqueued since ntru_scratch_acquire (records the event the next other-stream user waits for).
NTRU_HIDDEN int ntru_scratch_release(ntru_engine *eng);

// Scope guard around the two calls above: acquires now, releases on EVERY way out of the caller (an early error return must still
// record the event: kernels that are already enqueued keep using the buffer, and the next call on another stream has to wait for them).
struct ScratchHold {
  ntru_engine *eng;
  char *p = nullptr;
  int rc;
  ScratchHold(ntru_engine *e, size_t bytes) : eng(e) { rc = ntru_scratch_acquire(e, bytes, &p); }
  ~ScratchHold() { if (rc == NTRU_OK) (void)ntru_scratch_release(eng); }
  ScratchHold(const ScratchHold &) = delete;
  ScratchHold &operator=(const ScratchHold &) = delete;
};

// ---- host helpers shared by the translation units ---------------------------------------------------------------------------
#define NTRU_NOT_TAKEN (-1000)      // a launcher's "not my parameter range"; never leaves the library

static inline int fail(int code, const std::string &msg) { return ntru_fail(code, msg); }
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
// Workgroups of 256 for an elementwise kernel.  Kernels that move 16 bytes per lane and access (streaming = true) get TWO per CU: more
// resident waves lower the HBM rate (the add of two ciphertext batches: 5.0 TB/s with 8 per CU, 5.9 with 2, 4.8 with 1:
// profiles/archive/r03_ab_elementwise_grid.txt); element-per-lane kernels keep 8.  NTRU_EW_PER_CU overrides the streaming figure (experiments).
static inline dim3 elementwise_grid(const ntru_engine *eng, long total, bool streaming = false) {
  static const int ew_env = getenv("NTRU_EW_PER_CU") ? atoi(getenv("NTRU_EW_PER_CU")) : 0;
  const int per_cu = streaming ? (ew_env > 0 ? ew_env : 2) : 8;
  long blocks = (total + 255) / 256, cap = (long)eng->cus * per_cu;
  return dim3((unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks)));
}
// (N, q, B) of the packed kernels: q a power of two <= 65536, 2 <= N <= NTRU_MAX_N
NTRU_HIDDEN int ntru_check_common(const ntru_engine *eng, int N, int q, long B);

// ---- launchers (defined next to their kernels) --------------------------------------------------------------------------------
// encryptBits / decryptBits with a shared key on the matrix cores; rows at a pitch of ld elements
NTRU_HIDDEN int ntru_launch_encrypt_matrix(ntru_engine *eng, int N, int q, int ld, const uint16_t *d_h, const uint8_t *d_r,
                                           const uint8_t *d_m, int64_t B, uint16_t *d_e, uint16_t *d_quotE);
NTRU_HIDDEN int ntru_launch_decrypt_matrix(ntru_engine *eng, int N, int q, int p, int ld, const int8_t *d_f, const uint8_t *d_fp,
                                           const uint16_t *d_e, int64_t B, uint8_t *d_value, uint16_t *d_quot1, uint16_t *d_rem1,
                                           uint8_t *d_quot2);
// decryptBits + packOutput(p - 1, N, value) in one kernel; d_value may be NULL (NTRU_NOT_TAKEN outside the matrix path's range)
NTRU_HIDDEN int ntru_launch_decrypt_pack_matrix(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const uint8_t *d_fp,
                                                const uint16_t *d_e, int64_t B, uint8_t *d_value, uint64_t *d_packed, int out_size);
// the row-image variants (matrix_rowimage.hip): dense rows, one eight-wave workgroup per CU, results leave through LDS images
NTRU_HIDDEN int ntru_launch_encrypt_rowimage(ntru_engine *eng, int N, int q, int ld, const uint16_t *d_h, const uint8_t *d_r,
                                             const uint8_t *d_m, int64_t B, uint16_t *d_e, uint16_t *d_quotE);
NTRU_HIDDEN int ntru_launch_encrypt_pack_rowimage(ntru_engine *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m,
                                                  int64_t B, uint64_t *d_packed, int os);
// the same on the vector-ALU families (dense rows); never NTRU_NOT_TAKEN
NTRU_HIDDEN int ntru_launch_encrypt_valu(ntru_engine *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m,
                                         int64_t B, uint16_t *d_e, uint16_t *d_quotE);
NTRU_HIDDEN int ntru_launch_decrypt_valu(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const uint8_t *d_fp,
                                         const uint16_t *d_e, int64_t B, uint8_t *d_value, uint16_t *d_quot1, uint16_t *d_rem1,
                                         uint8_t *d_quot2);
// per-item products
// one Newton round of the key inversion (v from kb to m bits) as one kernel; _applies: kb <= 7, kb < m <= 2 kb, per-item matrix range
NTRU_HIDDEN bool ntru_newton_round_matrix_applies(const ntru_engine *eng, int N, int kb, int m);
NTRU_HIDDEN int ntru_launch_newton_round_matrix(ntru_engine *eng, int N, int kb, int m, const int8_t *d_f, uint16_t *d_v, long B);
NTRU_HIDDEN int ntru_launch_polymul_matrix(ntru_engine *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b, int64_t B,
                                           uint16_t *d_quot, uint16_t *d_rem);
NTRU_HIDDEN int ntru_launch_polymul_valu(ntru_engine *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b, int64_t B,
                                         uint16_t *d_quot, uint16_t *d_rem);
NTRU_HIDDEN bool ntru_product_tern_matrix_applies(const ntru_engine *eng, int N, int q);
// ((mul * a) mod q) * s with s ternary per item; d_quot may be NULL
NTRU_HIDDEN int ntru_launch_product_tern_matrix(ntru_engine *eng, int N, int q, uint32_t mul, const uint16_t *d_a, const int8_t *d_s,
                                                long B, uint16_t *d_rem);
NTRU_HIDDEN int ntru_launch_public_key_valu(ntru_engine *eng, int N, int q, int p, const uint16_t *d_fq, const int8_t *d_g, int64_t B,
                                            uint16_t *d_h);
NTRU_HIDDEN int ntru_launch_verify_keys_matrix(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const int8_t *d_g,
                                               const uint16_t *d_fq, const uint8_t *d_fp, const uint16_t *d_h, int64_t B,
                                               uint16_t *d_quot_fq, uint16_t *d_rem_fq, uint8_t *d_quot_fp, uint8_t *d_rem_fp,
                                               uint16_t *d_quot_h, uint16_t *d_rem_h, uint8_t *d_flags);
NTRU_HIDDEN int ntru_launch_verify_keys_valu(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const int8_t *d_g,
                                             const uint16_t *d_fq, const uint8_t *d_fp, const uint16_t *d_h, int64_t B,
                                             uint16_t *d_quot_fq, uint16_t *d_rem_fq, uint8_t *d_quot_fp, uint8_t *d_rem_fp,
                                             uint16_t *d_quot_h, uint16_t *d_rem_h, uint8_t *d_flags);

#endif
