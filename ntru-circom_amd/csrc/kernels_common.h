// kernels_common.h -- what the kernel translation units of libntru_engine.so share: scalar types, the workgroup shape, small device
// helpers, and the host-side launch helpers (persistent grid from the occupancy query).
#ifndef NTRU_KERNELS_COMMON_H
#define NTRU_KERNELS_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>

#include "engine_internal.h"

typedef unsigned short u16;
typedef unsigned int u32;
typedef u16 u16x2 __attribute__((ext_vector_type(2)));

#define WAVES_PER_BLOCK 4
#define BLOCK_THREADS (WAVES_PER_BLOCK * 64)

static __device__ __forceinline__ u16x2 as_pair(u32 v) { return __builtin_bit_cast(u16x2, v); }
static __device__ __forceinline__ u32 as_u32(u16x2 v) { return __builtin_bit_cast(u32, v); }

// x mod a small runtime modulus; p = 3 (every NTRU parameter set) gets the constant-divisor sequence.
static __device__ __forceinline__ u32 mod_small(u32 x, u32 m) { return m == 3u ? x % 3u : x % m; }

// Order this wave's LDS writes before its later LDS reads (regions touched here are private to one wave).
static __device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// Persistent grid: as many workgroups as are co-resident (occupancy query for this kernel and LDS size) x CUs, capped
// by the work available.  A grid larger than residency would run its tail at a fraction of the chip.  The dynamic-LDS limit of
// a kernel is raised at its first use, inside ntru_blocks_per_cu.
template <class Kern>
static int resident_grid(const ntru_engine *eng, Kern kern, size_t lds, long work_blocks, dim3 *grid, int threads = BLOCK_THREADS) {
  int per_cu = 0;
  if (int rc = ntru_blocks_per_cu(const_cast<ntru_engine *>(eng), (const void *)kern, threads, lds, &per_cu)) return rc;
  if (eng->max_blocks_per_cu && eng->max_blocks_per_cu < per_cu) per_cu = eng->max_blocks_per_cu;
  long blocks = (long)eng->cus * per_cu;
  if (blocks > work_blocks) blocks = work_blocks;
  *grid = dim3((unsigned)(blocks < 1 ? 1 : blocks));
  return NTRU_OK;
}

static inline void note_kernel(ntru_engine *eng, const char *family, int K, int me) {
  if (me >= 0) snprintf(eng->last_kernel, sizeof eng->last_kernel, "%s<%d,%d>", family, K, me);
  else snprintf(eng->last_kernel, sizeof eng->last_kernel, "%s<%d>", family, K);
}

#endif
