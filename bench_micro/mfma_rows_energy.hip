// What does a v_mfma_i32_32x32x32_i8 cost when only SOME of its 32 A rows are non-zero?  The per-item products (csrc/matrix_peritem.hip)
// issue 2 NT instructions per plane of which on average 41 % of the rows carry data (the rest are the zero rows of a shifted chunk
// matrix): if the instruction's energy followed the issued MACs, a 16-row tile (56 % useful) would save energy; if it follows the
// non-zero products, it would not.  Every SIMD issues the instruction back to back on four accumulators for `seconds`; A = random
// bytes in the first `rows` rows (lanes r < rows of both half-waves), zero elsewhere; B = random bytes.  Run under
// tools/power_sample.py, which reads the socket power meanwhile:
//   hipcc -O3 --offload-arch=gfx950 -o mfma_rows_energy mfma_rows_energy.hip
//   python3 tools/power_sample.py -- bench_micro/mfma_rows_energy <rows 0..32> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_spin(unsigned long long ticks, int rows, const int *__restrict__ in, unsigned long long *out) {
  const int lane = threadIdx.x & 63;
  v4i a = {in[4 * lane], in[4 * lane + 1], in[4 * lane + 2], in[4 * lane + 3]};
  const v4i b = {in[256 + 4 * lane], in[257 + 4 * lane], in[258 + 4 * lane], in[259 + 4 * lane]};
  if ((lane & 31) >= rows) a = (v4i){0, 0, 0, 0};
  v16i acc[4];
  for (int t = 0; t < 4; t++) for (int g = 0; g < 16; g++) acc[t][g] = 0;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  unsigned long long it = 0, r1 = r0;
  while (r1 - r0 < ticks) {
    for (int k = 0; k < 64; k++) {
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
    }
    it += 256;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int t = 0; t < 4; t++) for (int g = 0; g < 16; g++) s ^= acc[t][g];
  if (lane == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    out[3 * w] = it + (s == 0x7fffffff); out[3 * w + 1] = t1 - t0; out[3 * w + 2] = r1 - r0;
  }
}

int main(int argc, char **argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 32;
  const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  char pci[64]; CK(hipDeviceGetPCIBusId(pci, sizeof pci, 0));
  printf("pci_bus_id %s\n", pci);
  std::vector<int> h(512);
  unsigned x = 7; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (int)x; }
  int *d_in; unsigned long long *d_out;
  const int blocks = prop.multiProcessorCount;               // one four-wave workgroup per CU: one wave per SIMD
  CK(hipMalloc(&d_in, 2048)); CK(hipMemcpy(d_in, h.data(), 2048, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, (size_t)blocks * 4 * 3 * 8));
  hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, 0, (unsigned long long)(seconds * 1e8), rows, d_in, d_out);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> o((size_t)blocks * 4 * 3);
  CK(hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost));
  double instr = 0, clk = 0, ref = 0;
  for (int w = 0; w < blocks * 4; w++) { instr += (double)o[3 * w]; clk += (double)o[3 * w + 1]; ref += (double)o[3 * w + 2]; }
  printf("rows %d: %.3e matrix instructions in %.2f s = %.2f G/s; shader clock %.3f GHz\n", rows, instr, ref / (blocks * 4) / 1e8,
         instr / (ref / (blocks * 4) / 1e8) / 1e9, clk / ref * 0.1);
  return 0;
}
