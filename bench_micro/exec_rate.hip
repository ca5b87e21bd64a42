// Micro-benchmark: does gfx950 skip the inactive 32-lane pass of a wave64 VALU op?  Plus rates of helper VALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int NACC = 16, REPS = 8;

template <int MODE>   // 0 full exec, 1 low half, 2 high half, 3 zero, 4 alternating halves per instruction group
__global__ void k_exec(const unsigned* in, unsigned* out, int iters) {
  unsigned acc[NACC];
  for (int t = 0; t < NACC; t++) acc[t] = threadIdx.x + t;
  unsigned a = in[threadIdx.x & 63];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REPS; r++) {
      if (MODE == 1) asm volatile("s_mov_b64 exec, 0x00000000ffffffff" ::: "exec");
      if (MODE == 2) asm volatile("s_mov_b32 exec_lo, 0\n\ts_mov_b32 exec_hi, -1" ::: "exec");
      if (MODE == 3) asm volatile("s_mov_b64 exec, 0" ::: "exec");
      if (MODE == 4) { if (r & 1) asm volatile("s_mov_b32 exec_lo, 0\n\ts_mov_b32 exec_hi, -1" ::: "exec"); else asm volatile("s_mov_b64 exec, 0x00000000ffffffff" ::: "exec"); }
#pragma unroll
      for (int t = 0; t < NACC; t++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[t]) : "v"(a));
      if (MODE != 0) asm volatile("s_mov_b64 exec, -1" ::: "exec");
    }
  }
  unsigned s = 0;
  for (int t = 0; t < NACC; t++) s ^= acc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define OPK(NAME, ASM)                                                                         \
  __global__ void NAME(const unsigned* in, unsigned* out, int iters) {                          \
    unsigned acc[NACC];                                                                         \
    for (int t = 0; t < NACC; t++) acc[t] = threadIdx.x + t;                                    \
    unsigned a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];                         \
    for (int it = 0; it < iters; it++) {                                                        \
      _Pragma("unroll") for (int r = 0; r < REPS; r++) {                                        \
        _Pragma("unroll") for (int t = 0; t < NACC; t++) asm volatile(ASM : "+v"(acc[t]) : "v"(a), "v"(b)); \
      }                                                                                         \
    }                                                                                           \
    unsigned s = 0;                                                                             \
    for (int t = 0; t < NACC; t++) s ^= acc[t];                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                             \
  }
OPK(k_alignbit, "v_alignbit_b32 %0, %1, %0, %2")
OPK(k_alignbit_c, "v_alignbit_b32 %0, %1, %0, 16")
OPK(k_lshl_add, "v_lshl_add_u32 %0, %1, 1, %0")
OPK(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
OPK(k_sub, "v_sub_u32 %0, %0, %1")
OPK(k_and, "v_and_b32 %0, %1, %0")
OPK(k_mov, "v_mov_b32 %0, %1")
OPK(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OPK(k_perm, "v_perm_b32 %0, %1, %0, %2")
OPK(k_add3, "v_add3_u32 %0, %0, %1, %2")
OPK(k_xad, "v_xad_u32 %0, %0, %1, %2")
OPK(k_bfe, "v_bfe_u32 %0, %0, %1, %2")
OPK(k_lshlrev, "v_lshlrev_b32 %0, 3, %0")
OPK(k_mul_u24, "v_mul_u32_u24 %0, %1, %0")
OPK(k_mul_lo, "v_mul_lo_u32 %0, %1, %0")

typedef void (*kern_t)(const unsigned*, unsigned*, int);
int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  unsigned *in, *out;
  CK(hipMalloc(&in, 4096)); CK(hipMemset(in, 1, 4096));
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct C { const char* name; kern_t k; } cs[] = {
    {"v_add_u32 exec=full", k_exec<0>}, {"v_add_u32 exec=low32", k_exec<1>}, {"v_add_u32 exec=high32", k_exec<2>},
    {"v_add_u32 exec=0", k_exec<3>}, {"v_add_u32 exec alt halves", k_exec<4>},
    {"v_alignbit_b32 (vgpr sh)", k_alignbit}, {"v_alignbit_b32 (const sh)", k_alignbit_c}, {"v_lshl_add_u32", k_lshl_add},
    {"v_and_or_b32", k_and_or}, {"v_sub_u32", k_sub}, {"v_and_b32", k_and}, {"v_mov_b32", k_mov}, {"v_cndmask_b32", k_cndmask},
    {"v_perm_b32", k_perm}, {"v_add3_u32", k_add3}, {"v_xad_u32", k_xad}, {"v_bfe_u32", k_bfe}, {"v_lshlrev_b32", k_lshlrev},
    {"v_mul_u32_u24", k_mul_u24}, {"v_mul_lo_u32", k_mul_lo},
  };
  const int iters = 2000;
  for (auto& c : cs) {
    for (int wps : {2, 8}) {
      int blocks = cus * wps;
      hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, in, out, 10);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, in, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double winstr = (double)blocks * 4 * iters * NACC * REPS;           // wave-instructions
      double cyc = best * 1e-3 * 2.4e9 * (cus * 4) / winstr;                // SIMD cycles per wave-instruction
      printf("%-28s waves/SIMD=%d  %8.3f ms  %.2f SIMD-cycles per wave-instr\n", c.name, wps, best, cyc);
    }
  }
  return 0;
}
