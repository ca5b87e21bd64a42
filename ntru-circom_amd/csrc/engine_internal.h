// engine_internal.h -- what the translation units of libntru_engine.so share (not part of the C ABI).
//
//   ntru_engine.hip   kernels of the hot path + the *_dev entry points + engine life cycle
//   ntru_host.hip     host-pointer entry points: pinned staging, two streams, chunked H2D / kernel / D2H pipeline
//   ntru_generic.hip  reference-faithful generic family (arbitrary divisors, moduli up to 2^26, signed coefficients)
#ifndef NTRU_ENGINE_INTERNAL_H
#define NTRU_ENGINE_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
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

// Host-path staging (ntru_host.hip): two slots, each with a pinned host arena, a device arena, its own stream and an
// event that marks "slot free again".  Chunk k uses slot k & 1, so the H2D copy of chunk k+1 and the D2H copy of
// chunk k-1 run while chunk k computes.
struct HostSlot {
  hipStream_t stream = nullptr;
  GrowBuf pinned;           // hipHostMalloc
  GrowBuf dev;              // hipMalloc
  GrowBuf scratch;          // temporaries of a *_dev call issued on this slot's stream
  bool busy = false;        // work enqueued on `stream` that has not been waited for yet
};

struct ntru_engine {
  int device;
  hipStream_t stream;       // caller's stream for the *_dev entry points (never owned)
  int cus;
  int path;                 // 0 auto, 1 MAC kernels, 2 add path, 3 add path without dot8, 4 matrix-core path
  int max_blocks_per_cu;    // NTRU_MAX_BLOCKS_PER_CU read once at creation (tuning experiments); 0 = no cap
  char last_kernel[64];     // name of the kernel the last *_dev call launched (reporting only)
  HostSlot slot[2];
  GrowBuf shared_dev;       // shared key rows of the host path (h, f, fp)
  GrowBuf scratch_dev;      // Newton temporaries of ntru_invert_key_batch_dev on the caller's stream
  GrowBuf *cur_scratch;     // = &scratch_dev, or a slot's scratch while the host path borrows the engine for that slot
  // (kernel function, LDS bytes, block size) -> co-resident blocks per CU, filled at first use
  struct OccEntry { const void *fn; size_t lds; int threads; int per_cu; };
  OccEntry occ[64];
  int n_occ;
};

// Grows `b` to at least `bytes` of device memory (contents are not preserved).
NTRU_HIDDEN int ntru_grow_dev(GrowBuf *b, size_t bytes);
// Grows `b` to at least `bytes` of pinned host memory.
NTRU_HIDDEN int ntru_grow_pinned(GrowBuf *b, size_t bytes);
// hipOccupancyMaxActiveBlocksPerMultiprocessor, cached per engine.
NTRU_HIDDEN int ntru_blocks_per_cu(ntru_engine *eng, const void *fn, int threads, size_t lds, int *per_cu);

#endif
