// Micro-benchmark of the ternary add-path step structure: per step two bit tests + branches on wave-uniform masks and
// K full-rate v_add_u32 into one of two accumulator sets.  Calibrates the scalar-issue vs VALU-issue balance on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o step_rate step_rate.hip && ./step_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int K> struct Ops;
#define ADD(i) "v_add_u32 %[s" #i "], %[s" #i "], %[w" #i "]\n\t"
#define SO(i) [s##i] "+v"(S[i])
#define WI(i) [w##i] "v"(W[i])
template <> struct Ops<7> {
  template <int BIT> static __device__ __forceinline__ void add_if(unsigned (&S)[7], const unsigned (&W)[7], unsigned m) {
    asm volatile("s_bitcmp1_b32 %[m], %[b]\n\ts_cbranch_scc0 1f\n\t" ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) "1:\n"
                 : SO(0), SO(1), SO(2), SO(3), SO(4), SO(5), SO(6)
                 : [m] "s"(m), [b] "i"(BIT), WI(0), WI(1), WI(2), WI(3), WI(4), WI(5), WI(6) : "scc");
  }
};
template <> struct Ops<13> {
  template <int BIT> static __device__ __forceinline__ void add_if(unsigned (&S)[13], const unsigned (&W)[13], unsigned m) {
    asm volatile("s_bitcmp1_b32 %[m], %[b]\n\ts_cbranch_scc0 1f\n\t" ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6)
                 ADD(7) ADD(8) ADD(9) ADD(10) ADD(11) ADD(12) "1:\n"
                 : SO(0), SO(1), SO(2), SO(3), SO(4), SO(5), SO(6), SO(7), SO(8), SO(9), SO(10), SO(11), SO(12)
                 : [m] "s"(m), [b] "i"(BIT), WI(0), WI(1), WI(2), WI(3), WI(4), WI(5), WI(6), WI(7), WI(8), WI(9), WI(10),
                   WI(11), WI(12) : "scc");
  }
};

#define ADD1(i) "v_add_u32 %[a" #i "], %[a" #i "], %[w" #i "]\n\t"
#define ADD2(i) "v_add_u32 %[b" #i "], %[b" #i "], %[w" #i "]\n\t"
template <int K> struct Nest;
template <> struct Nest<7> {
  template <int BIT> static __device__ __forceinline__ void step(unsigned (&S1)[7], unsigned (&S2)[7], const unsigned (&W)[7], unsigned ones, unsigned twos) {
    asm volatile("s_bitcmp1_b32 %[o], %[b]\n\ts_cbranch_scc1 2f\n\ts_bitcmp1_b32 %[t], %[b]\n\ts_cbranch_scc0 3f\n\t" ADD2(0) ADD2(1) ADD2(2) ADD2(3) ADD2(4) ADD2(5) ADD2(6) "s_branch 3f\n2:\n\t" ADD1(0) ADD1(1) ADD1(2) ADD1(3) ADD1(4) ADD1(5) ADD1(6) "3:\n"
                 : [a0] "+v"(S1[0]), [a1] "+v"(S1[1]), [a2] "+v"(S1[2]), [a3] "+v"(S1[3]), [a4] "+v"(S1[4]), [a5] "+v"(S1[5]), [a6] "+v"(S1[6]), [b0] "+v"(S2[0]), [b1] "+v"(S2[1]), [b2] "+v"(S2[2]), [b3] "+v"(S2[3]), [b4] "+v"(S2[4]), [b5] "+v"(S2[5]), [b6] "+v"(S2[6])
                 : [o] "s"(ones), [t] "s"(twos), [b] "i"(BIT), WI(0), WI(1), WI(2), WI(3), WI(4), WI(5), WI(6) : "scc");
  }
};
template <> struct Nest<13> {
  template <int BIT> static __device__ __forceinline__ void step(unsigned (&S1)[13], unsigned (&S2)[13], const unsigned (&W)[13], unsigned ones, unsigned twos) {
    asm volatile("s_bitcmp1_b32 %[o], %[b]\n\ts_cbranch_scc1 2f\n\ts_bitcmp1_b32 %[t], %[b]\n\ts_cbranch_scc0 3f\n\t" ADD2(0) ADD2(1) ADD2(2) ADD2(3) ADD2(4) ADD2(5) ADD2(6) ADD2(7) ADD2(8) ADD2(9) ADD2(10) ADD2(11) ADD2(12) "s_branch 3f\n2:\n\t" ADD1(0) ADD1(1) ADD1(2) ADD1(3) ADD1(4) ADD1(5) ADD1(6) ADD1(7) ADD1(8) ADD1(9) ADD1(10) ADD1(11) ADD1(12) "3:\n"
                 : [a0] "+v"(S1[0]), [a1] "+v"(S1[1]), [a2] "+v"(S1[2]), [a3] "+v"(S1[3]), [a4] "+v"(S1[4]), [a5] "+v"(S1[5]), [a6] "+v"(S1[6]), [a7] "+v"(S1[7]), [a8] "+v"(S1[8]), [a9] "+v"(S1[9]), [a10] "+v"(S1[10]), [a11] "+v"(S1[11]), [a12] "+v"(S1[12]), [b0] "+v"(S2[0]), [b1] "+v"(S2[1]), [b2] "+v"(S2[2]), [b3] "+v"(S2[3]), [b4] "+v"(S2[4]), [b5] "+v"(S2[5]), [b6] "+v"(S2[6]), [b7] "+v"(S2[7]), [b8] "+v"(S2[8]), [b9] "+v"(S2[9]), [b10] "+v"(S2[10]), [b11] "+v"(S2[11]), [b12] "+v"(S2[12])
                 : [o] "s"(ones), [t] "s"(twos), [b] "i"(BIT), WI(0), WI(1), WI(2), WI(3), WI(4), WI(5), WI(6), WI(7), WI(8), WI(9), WI(10), WI(11), WI(12) : "scc");
  }
};

template <int K, int J>
__device__ __forceinline__ void nsteps(unsigned (&S1)[K], unsigned (&S2)[K], unsigned (&W)[K], unsigned ones, unsigned twos) {
  if constexpr (J < 2 * K) {
    Nest<K>::template step<J>(S1, S2, W, ones, twos);
    W[J % K] += 1;
    nsteps<K, J + 1>(S1, S2, W, ones, twos);
  }
}
template <int K>
__global__ void kn(const unsigned* masks, unsigned* out, int nblk, int iters) {
  unsigned S1[K], S2[K], W[K];
  for (int t = 0; t < K; t++) { S1[t] = threadIdx.x + t; S2[t] = t; W[t] = threadIdx.x * 7 + t; }
  for (int it = 0; it < iters; it++)
    for (int m = 0; m < nblk; m++) {
      unsigned ones = __builtin_amdgcn_readfirstlane(masks[2 * m]), twos = __builtin_amdgcn_readfirstlane(masks[2 * m + 1]);
      nsteps<K, 0>(S1, S2, W, ones, twos);
    }
  unsigned s = 0;
  for (int t = 0; t < K; t++) s ^= S1[t] ^ S2[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K, int J>
__device__ __forceinline__ void steps(unsigned (&S1)[K], unsigned (&S2)[K], unsigned (&W)[K], unsigned ones, unsigned twos) {
  if constexpr (J < 2 * K) {
    Ops<K>::template add_if<J>(S1, W, ones);
    Ops<K>::template add_if<J>(S2, W, twos);
    W[J % K] += 1;                                  // stands for the window-slot refresh (one VALU per step)
    steps<K, J + 1>(S1, S2, W, ones, twos);
  }
}

template <int K>
__global__ void k(const unsigned* masks, unsigned* out, int nblk, int iters) {
  unsigned S1[K], S2[K], W[K];
  for (int t = 0; t < K; t++) { S1[t] = threadIdx.x + t; S2[t] = t; W[t] = threadIdx.x * 7 + t; }
  for (int it = 0; it < iters; it++)
    for (int m = 0; m < nblk; m++) {
      unsigned ones = __builtin_amdgcn_readfirstlane(masks[2 * m]), twos = __builtin_amdgcn_readfirstlane(masks[2 * m + 1]);
      steps<K, 0>(S1, S2, W, ones, twos);
    }
  unsigned s = 0;
  for (int t = 0; t < K; t++) s ^= S1[t] ^ S2[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, bool NESTED = false> void run(const char* name, int cus, unsigned* d_masks, unsigned* d_out, double frac_nonzero) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nblk = 64, iters = 40;
  for (int wps : {1, 2, 4, 6, 8}) {
    int blocks = cus * wps;
    if (NESTED) hipLaunchKernelGGL(kn<K>, dim3(blocks), dim3(256), 0, 0, d_masks, d_out, nblk, 2); else hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d_masks, d_out, nblk, 2);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      if (NESTED) hipLaunchKernelGGL(kn<K>, dim3(blocks), dim3(256), 0, 0, d_masks, d_out, nblk, iters); else hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d_masks, d_out, nblk, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double steps_per_wave = (double)nblk * iters * 2 * K;
    double cyc_per_step_simd = best * 1e-3 * 2.4e9 / (steps_per_wave * wps);   // SIMD cycles per wave-step
    double adds = steps_per_wave * frac_nonzero * K * (double)blocks * 256;
    printf("%-10s K=%2d waves/SIMD=%d  %.3f ms  %.1f SIMD-cycles per wave-step  %.1f T lane-adds/s\n", name, K, wps, best,
           cyc_per_step_simd, adds / (best * 1e-3) / 1e12);
  }
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  std::vector<unsigned> h(128);
  unsigned x = 12345; int nz = 0, tot = 0;
  for (int m = 0; m < 64; m++) {
    unsigned ones = 0, twos = 0;
    for (int j = 0; j < 26; j++) { x = x * 1664525u + 1013904223u; unsigned c = (x >> 16) % 3; if (c == 1) ones |= 1u << j; if (c == 2) twos |= 1u << j; if (j < 26) { tot++; nz += c != 0; } }
    h[2 * m] = ones; h[2 * m + 1] = twos;
  }
  unsigned *d_masks, *d_out;
  CK(hipMalloc(&d_masks, 512)); CK(hipMemcpy(d_masks, h.data(), 512, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
  double f = (double)nz / tot;
  printf("nonzero fraction %.3f\n", f);
  run<7>("random", cus, d_masks, d_out, f);
  run<13>("random", cus, d_masks, d_out, f);
  run<7, true>("rand-nest", cus, d_masks, d_out, f);
  run<13, true>("rand-nest", cus, d_masks, d_out, f);
  for (auto& v : h) v = 0;
  CK(hipMemcpy(d_masks, h.data(), 512, hipMemcpyHostToDevice));
  run<7>("all-zero", cus, d_masks, d_out, 0);
  run<13>("all-zero", cus, d_masks, d_out, 0);
  for (int m = 0; m < 64; m++) { h[2 * m] = 0x3FFFFFF; h[2 * m + 1] = 0; }
  CK(hipMemcpy(d_masks, h.data(), 512, hipMemcpyHostToDevice));
  run<7>("all-ones", cus, d_masks, d_out, 1);
  run<13>("all-ones", cus, d_masks, d_out, 1);
  return 0;
}
