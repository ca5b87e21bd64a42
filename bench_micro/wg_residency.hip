// Micro-benchmark: how many SMALL workgroups does a CU really hold at once, and on which SIMDs do their waves land?
// Every workgroup records HW_ID / XCC_ID and its start / end time (constant 100 MHz clock) around a fixed spin; the host counts, per CU,
// the workgroups whose intervals overlap the middle of the first arrivals' interval and the waves per SIMD among them.
//   wg_residency <threads per workgroup> <workgroups per CU asked for> <LDS bytes per workgroup>
// hipcc -O3 --offload-arch=gfx950 -o wg_residency wg_residency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

struct Rec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };

__global__ void k(Rec* out, long spin_ticks) {
  extern __shared__ unsigned char lds[];
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds[0] = 1;
    while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Rec r{hw, xcc, t0, __builtin_amdgcn_s_memrealtime()};
    out[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
  }
}

int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 64, per_cu = argc > 2 ? atoi(argv[2]) : 12;
  const size_t lds = argc > 3 ? (size_t)atol(argv[3]) : 13312;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * per_cu, waves = threads / 64;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, threads, lds));
  Rec* d; CK(hipMalloc(&d, sizeof(Rec) * blocks * waves));
  std::vector<Rec> h((size_t)blocks * waves);
  for (int rep = 0; rep < 2; rep++) {                     // the second launch is the measured one
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d, 20000L);      // 200 us per workgroup
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost));
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& r : h) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
  // CU key: xcc (4 bits) | se_id (HW_ID bits 13-15) | sh (12) | cu_id (8-11); simd = bits 4-5
  std::map<unsigned, std::vector<const Rec*>> by_cu;
  for (auto& r : h) by_cu[((r.xcc_id & 15) << 16) | (r.hw_id & 0xFF00)].push_back(&r);
  long hist_res[64] = {0}; long simd_hist[4][16] = {{0}};
  for (auto& kv : by_cu) {
    const unsigned long long probe = tmin + 10000;        // 100 us after the first start: inside the first arrivals' spin
    int resident = 0, simd[4] = {0, 0, 0, 0};
    for (auto* r : kv.second) if (r->t0 <= probe && probe < r->t1) { resident++; simd[(r->hw_id >> 4) & 3]++; }
    hist_res[std::min(resident, 63)]++;
    for (int s = 0; s < 4; s++) simd_hist[s][std::min(simd[s], 15)]++;
  }
  printf("threads %d, asked %d workgroups per CU (%d waves), LDS %zu B, occupancy query %d; %zu CUs seen; whole launch %.0f us (one spin = 200 us)\n",
         threads, per_cu, per_cu * waves, lds, occ, by_cu.size(), (tmax - tmin) / 100.0);
  printf("  waves resident per CU at t = 100 us  (count of CUs):");
  for (int i = 0; i < 64; i++) if (hist_res[i]) printf("  %d waves: %ld", i, hist_res[i]);
  printf("\n");
  for (int s = 0; s < 4; s++) {
    printf("  SIMD %d waves (count of CUs):", s);
    for (int i = 0; i < 16; i++) if (simd_hist[s][i]) printf("  %d: %ld", i, simd_hist[s][i]);
    printf("\n");
  }
  return 0;
}
