// Micro-benchmark: do a matrix-heavy wave and a VALU-heavy wave on the SAME SIMD overlap?
// One workgroup of 512 threads per CU = 2 waves per SIMD (waves w and w+4 share SIMD w on gfx950).
// mode 0: all 8 waves MFMA loop;  1: all VALU loop;  2: waves 0-3 MFMA, waves 4-7 VALU;  3: waves 0-3 MFMA, 4-7 idle;
// 4: waves 0-3 idle, 4-7 VALU.
// hipcc -O3 --offload-arch=gfx950 -o coexec coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(const int* in, int* out, int iters, int mode) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
  const bool do_valu = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
  int res = 0;
  if (do_mfma) {
    v4i a = {in[threadIdx.x & 63], in[1], in[2], in[3]}, b = {in[4], in[threadIdx.x & 31], in[6], in[7]};
    v16i acc[4];
    for (int t = 0; t < 4; t++) for (int g = 0; g < 16; g++) acc[t][g] = t + g;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < 4; t++) for (int g = 0; g < 16; g++) res ^= acc[t][g];
  } else if (do_valu) {
    unsigned x[8];
    for (int t = 0; t < 8; t++) x[t] = in[t] + threadIdx.x;
    const unsigned c = in[9];
    for (int it = 0; it < iters; it++) {        // 32 independent-ish VALU ops per iteration (same count of "slots" as 4 MFMAs x 8 issue cycles)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int t = 0; t < 8; t++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[t]) : "v"(c));
    }
    for (int t = 0; t < 8; t++) res ^= x[t];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int *din, *dout; CK(hipMalloc(&din, 1024)); CK(hipMemset(din, 1, 1024)); CK(hipMalloc(&dout, (size_t)cus * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  const char* names[] = {"8 waves MFMA", "8 waves VALU", "4 MFMA + 4 VALU (paired on SIMDs)", "4 MFMA + 4 idle", "4 idle + 4 VALU"};
  for (int mode = 0; mode < 5; mode++) {
    k<<<cus, 512>>>(din, dout, 100, mode); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<<<cus, 512>>>(din, dout, iters, mode); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-36s %8.3f ms   (per iteration: %.1f cycles @2.4GHz; MFMA alone = 4 x 32 = 128, VALU alone = 32 x ~2.3-4)\n", names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
