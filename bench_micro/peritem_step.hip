// Probe for the per-item matrix loops (k_verify_keys_m and friends, csrc/matrix_peritem.hip): what ONE contraction step costs when
// the chunk rows (the MFMA A operand) are (a) read from LDS every step, as the kernels did until round 5, or (b) kept in registers and moved one row per
// step with ONE v_and_b32_dpp per dword (wave_shr / wave_shl + a lane mask that cuts the half-wave seam), and when NPL planes share
// one Toeplitz fragment read.  No HBM traffic, operands random bytes; the loop shape (26 low + 25 high steps + split diagonal,
// compile-time offsets) is the product's.  Prints SIMD clocks per step at 1-3 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o peritem_step peritem_step.hip && ./peritem_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned u32;

constexpr int NT = 26, TPITCH = ((16 * NT + 31) / 32) * 32 + 8, PAD = 32;
constexpr int FA_BYTES = 32 * (NT + 2 * PAD), T_BYTES = 16 * TPITCH;

static __device__ __forceinline__ v4i shr_and(v4i a, v4i m) {   // lane l takes lane l - 1 (lane 0: zero), ANDed with m
  v4i o;
#pragma unroll
  for (int c = 0; c < 4; c++) o[c] = __builtin_amdgcn_update_dpp(0, a[c], 0x138, 0xf, 0xf, true) & m[c];
  return o;
}
static __device__ __forceinline__ v4i shl_and(v4i a, v4i m) {   // lane l takes lane l + 1 (lane 63: zero)
  v4i o;
#pragma unroll
  for (int c = 0; c < 4; c++) o[c] = __builtin_amdgcn_update_dpp(0, a[c], 0x130, 0xf, 0xf, true) & m[c];
  return o;
}

// NPL planes share the fragment; the first NDPP of them are shifted in registers, the others read from LDS.
template <int NPL, int NDPP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_steps(int iters, const int *__restrict__ seed, int *__restrict__ out, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
  unsigned char *base = lds + (size_t)wave * lds_per_wave;
  u32 *T = (u32 *)base;
  unsigned char *fa = base + T_BYTES;
  for (int i = lane; i < lds_per_wave / 4; i += 64) ((u32 *)base)[i] = (u32)seed[(i + 64 * wave) & 1023];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int y0 = 32 * NT - 1 - r + 16 * hh;
  // ds_read2_b32 takes 8-bit dword offsets: one base for the low pass (d = NT - 1 at offset 0), one for the high pass (d = 0 at offset 0)
  const u32 *tb_low = T + (y0 & 3) * TPITCH + (y0 >> 2) - 8 * (NT - 1), *tb_high = tb_low + 8 * (NT - 1);
  const unsigned char *pa[3];
  for (int p = 0; p < 3; p++) pa[p] = fa + p * FA_BYTES + 32 * PAD + 32 * r + 16 * hh - 32 * (NT - 1);
  v4i mup, mdn;
  for (int c = 0; c < 4; c++) { mup[c] = lane == 32 ? 0 : -1; mdn[c] = lane == 31 ? 0 : -1; asm volatile("" : "+v"(mup[c]), "+v"(mdn[c])); }
  u32 mlow[4];
  for (int c = 0; c < 4; c++) {
    u32 mk = 0;
    for (int jj = 0; jj < 4; jj++) mk |= (r >= 16 * hh + 4 * c + jj) ? (0xFFu << (8 * jj)) : 0u;
    mlow[c] = mk;
  }
  v16i L[NPL], H[NPL];
  for (int p = 0; p < NPL; p++) for (int i = 0; i < 16; i++) { L[p][i] = 0; H[p][i] = 0; }
  auto frag = [&](int d) {                                 // d compile-time after unrolling: immediate offsets
    const u32 *p = d >= 0 ? tb_low + 8 * (NT - 1 - d) : tb_high + 8 * (-d);
    return (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
  };
  auto rows = [&](int p, int d) { return *(const v4i *)(pa[p] + 32 * (NT - 1 - d)); };
  for (int it = 0; it < iters; it++) {
    v4i F[NPL], A[NPL];
#pragma unroll
    for (int p = 0; p < NPL; p++) { F[p] = rows(p, 0); A[p] = F[p]; }
    // diagonal
    {
      const v4i w = frag(0);
      v4i wl, wh;
#pragma unroll
      for (int c = 0; c < 4; c++) { wl[c] = (int)((u32)w[c] & mlow[c]); wh[c] = (int)((u32)w[c] & ~mlow[c]); }
#pragma unroll
      for (int p = 0; p < NPL; p++) {
        L[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], wl, L[p], 0, 0, 0);
        H[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], wh, H[p], 0, 0, 0);
      }
    }
    v4i wn = frag(1);
#pragma unroll
    for (int d = 1; d < NT; d++) {
      const v4i w = wn;
      wn = frag(d + 1 < NT ? d + 1 : -1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < NPL; p++) {
        A[p] = p < NDPP ? shr_and(A[p], mup) : rows(p, d);
        L[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], w, L[p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int p = 0; p < NPL; p++) A[p] = F[p];
#pragma unroll
    for (int d = -1; d > -NT; d--) {
      const v4i w = wn;
      wn = frag(d - 1 > -NT ? d - 1 : 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < NPL; p++) {
        A[p] = p < NDPP ? shl_and(A[p], mdn) : rows(p, d);
        H[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], w, H[p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  int s = 0;
  for (int p = 0; p < NPL; p++) for (int i = 0; i < 16; i++) s ^= L[p][i] ^ H[p][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- the same work on the 16-row tile (v_mfma_i32_16x16x64_i8: two distances per instruction, two 16-column halves) -------------------
// Output tiles in two row groups of 16 (NT = 26: rows 0-15, 16-25); a (group, distance pair) is issued only while it has rows in range:
// low part pairs (0,1) .. (24,25): group 0 for the first 8, group 1 for all 13; high part: group 0 for all 13, group 1 for the first 5
// -> 78 instructions of 16 clocks per plane instead of 52 of 32.  Lane (row, kq): kq 0,1 = the two K halves of distance d, kq 2,3 of
// distance d + 1 (one row further): a pair step moves every 16-lane row by TWO rows (row_shr:2); group 1 takes its entering rows
// from group 0's top two (row_ror:2 as the `old` operand): 3 lane shifts per dword and pair step, no seam mask.
typedef int v4i_ __attribute__((ext_vector_type(4)));
template <int NPL>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_steps16(int iters, const int *__restrict__ seed, int *__restrict__ out, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kq = lane >> 4;
  unsigned char *base = lds + (size_t)wave * lds_per_wave;
  u32 *T = (u32 *)base;
  unsigned char *fa = base + T_BYTES;
  for (int i = lane; i < lds_per_wave / 4; i += 64) ((u32 *)base)[i] = (u32)seed[(i + 64 * wave) & 1023];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  // fragment of distance pair starting at d, column half c: byte position as in the 32-wide scheme with column 16 c + col, K half kq & 1,
  // distance d + (kq >> 1)
  auto frag = [&](int d, int c) {
    const int y = 32 * NT - 1 - (16 * c + col) + 16 * (kq & 1) - 32 * (d + (kq >> 1)) + 32 * NT;      // (+32 NT: positive for every d in range)
    const u32 *p = T + (y & 3) * TPITCH + ((y >> 2) % (TPITCH - 4));
    return (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
  };
  auto rows0 = [&](int p, int g) { return *(const v4i *)(fa + p * FA_BYTES + 32 * PAD + 32 * (16 * g + col - (kq >> 1)) + 16 * (kq & 1)); };
  auto shift2 = [&](v4i &g0, v4i &g1) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int enter = __builtin_amdgcn_update_dpp(0, g0[c], 0x122, 0xf, 0xf, false);        // row_ror:2: lanes 0, 1 <- lanes 14, 15
      g1[c] = __builtin_amdgcn_update_dpp(enter, g1[c], 0x112, 0xf, 0xf, false);              // row_shr:2, lanes 0, 1 keep `enter`
      g0[c] = __builtin_amdgcn_update_dpp(0, g0[c], 0x112, 0xf, 0xf, true);                   // row_shr:2, zeros enter
    }
  };
  v4i acc[NPL][2][2][2];                                   // [plane][low / high][group][column half]
  for (int p = 0; p < NPL; p++) for (int h = 0; h < 2; h++) for (int g = 0; g < 2; g++) for (int c = 0; c < 2; c++) acc[p][h][g][c] = (v4i){0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {                          // low part, then high part (the shift direction does not matter to the probe)
      v4i A[NPL][2];
#pragma unroll
      for (int p = 0; p < NPL; p++) { A[p][0] = rows0(p, 0); A[p][1] = rows0(p, 1); }
      v4i w0 = frag(0, 0), w1 = frag(0, 1);
#pragma unroll
      for (int pr = 0; pr < 13; pr++) {
        const v4i c0 = w0, c1 = w1;
        w0 = frag(2 * pr + 2, 0); w1 = frag(2 * pr + 2, 1);
        const bool g0_on = h == 0 ? pr < 8 : true, g1_on = h == 0 ? true : pr < 5;
#pragma unroll
        for (int p = 0; p < NPL; p++) {
          if (g0_on) {
            acc[p][h][0][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p][0], c0, acc[p][h][0][0], 0, 0, 0);
            acc[p][h][0][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p][0], c1, acc[p][h][0][1], 0, 0, 0);
          }
          if (g1_on) {
            acc[p][h][1][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p][1], c0, acc[p][h][1][0], 0, 0, 0);
            acc[p][h][1][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p][1], c1, acc[p][h][1][1], 0, 0, 0);
          }
          shift2(A[p][0], A[p][1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  int s = 0;
  for (int p = 0; p < NPL; p++) for (int h = 0; h < 2; h++) for (int g = 0; g < 2; g++) for (int c = 0; c < 2; c++) for (int i = 0; i < 4; i++) s ^= acc[p][h][g][c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_sem(int *out) {
  const int lane = threadIdx.x;
  v4i a = {100 + lane, 0, 0, 0}, mup, mdn;
  for (int c = 0; c < 4; c++) { mup[c] = lane == 32 ? 0 : -1; mdn[c] = lane == 31 ? 0 : -1; }
  out[lane] = shr_and(a, mup)[0];
  out[64 + lane] = shl_and(a, mdn)[0];
}

template <int NPL, int NDPP>
static void run(int cus, const int *d_seed, int *d_out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int lds_per_wave = T_BYTES + 3 * FA_BYTES, iters = 200;
  CK(hipFuncSetAttribute((const void *)k_steps<NPL, NDPP>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * lds_per_wave));
  for (int wps : {1, 2, 3}) {
    const int blocks = cus * 2 * wps;                    // two-wave workgroups: 4 wps waves per CU
    hipLaunchKernelGGL((k_steps<NPL, NDPP>), dim3(blocks), dim3(128), 2 * lds_per_wave, 0, 2, d_seed, d_out, lds_per_wave);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_steps<NPL, NDPP>), dim3(blocks), dim3(128), 2 * lds_per_wave, 0, iters, d_seed, d_out, lds_per_wave);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double steps = (double)iters * (2 * NT);       // 2 NT matrix instructions per plane and pass pair
    printf("planes %d, in registers %d, waves/SIMD %d: %.3f ms  %.1f SIMD clocks (2.4 GHz) per step and wave = %.1f per matrix instruction; matrix pipe %.0f %%\n",
           NPL, NDPP, wps, best, best * 1e-3 * 2.4e9 / steps / wps, best * 1e-3 * 2.4e9 / steps / wps / NPL,
           100.0 * steps * wps * NPL * 32 / (best * 1e-3 * 2.4e9));
  }
}

template <class K>
static void sustain(const char *name, K kern, int cus, int wps, double seconds, const int *d_seed, int *d_out, double mfma_clocks_per_iter) {
  const int lds_per_wave = T_BYTES + 3 * FA_BYTES, iters = 200, blocks = cus * 2 * wps;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * lds_per_wave));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), 2 * lds_per_wave, 0, 2, d_seed, d_out, lds_per_wave);
  CK(hipDeviceSynchronize());
  double total_ms = 0; long n = 0;
  while (total_ms < seconds * 1e3) {
    CK(hipEventRecord(e0));
    for (int k = 0; k < 10; k++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(128), 2 * lds_per_wave, 0, iters, d_seed, d_out, lds_per_wave);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); total_ms += ms; n += 10;
  }
  const double ms = total_ms / n, products = (double)iters * blocks * 2;      // one "product set" (all planes) per iteration and wave
  printf("%s waves/SIMD %d: %.3f ms per launch = %.2f ns per product set and CU; matrix pipe %.0f %% (at 2.4 GHz)\n", name, wps, ms,
         ms * 1e6 / (products / cus), 100.0 * iters * wps * mfma_clocks_per_iter / (ms * 1e-3 * 2.4e9));
}

int main(int argc, char **argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<int> h(1024);
  unsigned x = 99; for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (int)x; }
  int *d_seed, *d_out;
  CK(hipMalloc(&d_seed, 4096)); CK(hipMemcpy(d_seed, h.data(), 4096, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, (size_t)cus * 8 * 128 * 4));
  if (argc > 1) {            // sustained mode for tools/power_sample.py: peritem_step <32|16> <waves per SIMD> <seconds>   (three planes, one fragment stream)
    char pci[64]; CK(hipDeviceGetPCIBusId(pci, sizeof pci, 0)); printf("pci_bus_id %s\n", pci); fflush(stdout);
    const int wps = argc > 2 ? atoi(argv[2]) : 2; const double sec = argc > 3 ? atof(argv[3]) : 3.0;
    if (atoi(argv[1]) == 16) sustain("16x16x64, rows in registers", k_steps16<3>, cus, wps, sec, d_seed, d_out, 3 * 78 * 16.0);
    else sustain("32x32x32, rows in registers", k_steps<3, 3>, cus, wps, sec, d_seed, d_out, 3 * 52 * 32.0);
    return 0;
  }
  {                                                        // semantics of the two shifts
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d_out);
    int o[128]; CK(hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int l = 0; l < 64; l++) {
      ok &= o[l] == ((l == 0 || l == 32) ? 0 : 100 + l - 1);
      ok &= o[64 + l] == ((l == 63 || l == 31) ? 0 : 100 + l + 1);
    }
    printf("wave_shr / wave_shl + seam mask: %s\n", ok ? "OK" : "MISMATCH");
  }
  run<2, 0>(cus, d_seed, d_out); run<2, 1>(cus, d_seed, d_out); run<2, 2>(cus, d_seed, d_out);
  run<3, 0>(cus, d_seed, d_out); run<3, 1>(cus, d_seed, d_out); run<3, 2>(cus, d_seed, d_out); run<3, 3>(cus, d_seed, d_out);
  run<1, 0>(cus, d_seed, d_out); run<1, 1>(cus, d_seed, d_out);
  return 0;
}
