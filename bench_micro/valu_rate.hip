// Micro-benchmark: issue rate of the candidate integer-MAC instructions on gfx950.
// Decides which instruction the schoolbook accumulate is built on ("Roofline").
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int NACC = 16;   // independent accumulators per lane
constexpr int REPS = 8;    // NACC*REPS instructions per loop trip

#define BODY(ASM, CSTR)                                                                     \
  unsigned acc[NACC];                                                                       \
  for (int t = 0; t < NACC; t++) acc[t] = threadIdx.x * 3 + t;                              \
  unsigned a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];                       \
  for (int it = 0; it < iters; it++) {                                                      \
    _Pragma("unroll") for (int r = 0; r < REPS; r++) {                                      \
      _Pragma("unroll") for (int t = 0; t < NACC; t++) asm volatile(ASM : "+v"(acc[t]) : "v"(a), CSTR(b)); \
    }                                                                                       \
  }                                                                                         \
  unsigned s = 0;                                                                           \
  for (int t = 0; t < NACC; t++) s ^= acc[t];                                               \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;

__global__ void k_pk_mad_u16(const unsigned* in, unsigned* out, int iters) { BODY("v_pk_mad_u16 %0, %1, %2, %0", "v") }
__global__ void k_pk_mad_u16_opsel(const unsigned* in, unsigned* out, int iters) { BODY("v_pk_mad_u16 %0, %1, %2, %0 op_sel_hi:[0,1,1]", "v") }
__global__ void k_pk_mad_u16_sgpr(const unsigned* in, unsigned* out, int iters) {
  unsigned acc[NACC];
  for (int t = 0; t < NACC; t++) acc[t] = threadIdx.x * 3 + t;
  unsigned a = in[threadIdx.x & 63];
  unsigned b = __builtin_amdgcn_readfirstlane(in[64]);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int t = 0; t < NACC; t++) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "s"(b));
    }
  }
  unsigned s = 0;
  for (int t = 0; t < NACC; t++) s ^= acc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pk_add_u16(const unsigned* in, unsigned* out, int iters) { BODY("v_pk_add_u16 %0, %1, %0", "v") (void)b; }
__global__ void k_pk_mul_lo_u16(const unsigned* in, unsigned* out, int iters) { BODY("v_pk_mul_lo_u16 %0, %1, %0", "v") (void)b; }
__global__ void k_mad_u32_u24(const unsigned* in, unsigned* out, int iters) { BODY("v_mad_u32_u24 %0, %1, %2, %0", "v") }
__global__ void k_mad_u16(const unsigned* in, unsigned* out, int iters) { BODY("v_mad_u16 %0, %1, %2, %0", "v") }
__global__ void k_add_u32(const unsigned* in, unsigned* out, int iters) { BODY("v_add_u32 %0, %1, %0", "v") (void)b; }
__global__ void k_dot4_u32_u8(const unsigned* in, unsigned* out, int iters) { BODY("v_dot4_u32_u8 %0, %1, %2, %0", "v") }
__global__ void k_dot8_u32_u4(const unsigned* in, unsigned* out, int iters) { BODY("v_dot8_u32_u4 %0, %1, %2, %0", "v") }
__global__ void k_dot2_u32_u16(const unsigned* in, unsigned* out, int iters) { BODY("v_dot2_u32_u16 %0, %1, %2, %0", "v") }
__global__ void k_dot4_i32_i8(const unsigned* in, unsigned* out, int iters) { BODY("v_dot4_i32_i8 %0, %1, %2, %0", "v") }
__global__ void k_fma_f32(const unsigned* in, unsigned* out, int iters) { BODY("v_fma_f32 %0, %1, %2, %0", "v") }
__global__ void k_pk_fma_f32(const unsigned* in, unsigned* out, int iters) {
  // 64-bit operands
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 acc[NACC];
  for (int t = 0; t < NACC; t++) acc[t] = (f2){(float)threadIdx.x, (float)t};
  f2 a = {(float)in[threadIdx.x & 63], 1.0f}, b = {(float)in[64 + (threadIdx.x & 63)], 0.5f};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REPS; r++) {
#pragma unroll
      for (int t = 0; t < NACC; t++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
    }
  }
  float s = 0;
  for (int t = 0; t < NACC; t++) s += acc[t].x + acc[t].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)s;
}

// LDS read rates in the access shapes the polymul kernel uses: per-lane stride-K dword reads and broadcast b128.
template <int STRIDE>
__global__ void k_lds_b32(const unsigned* in, unsigned* out, int iters) {
  __shared__ unsigned lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = in[i & 127];
  __syncthreads();
  unsigned s = 0;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int base = w * 2048 + lane * STRIDE;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((base + (it & 7)) * 4), "i"(r * 4));
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      s ^= v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(const unsigned*, unsigned*, int);

struct Case { const char* name; kern_t k; double macs_per_lane_instr; int instr_per_iter; };

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  unsigned* in; unsigned* out;
  CK(hipMalloc(&in, 4096 * 4));
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; i++) h[i] = 0x01020304u * (i + 1) + 0x00010001u;
  CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)cus * 32 * 64 * 4 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  Case cases[] = {
    {"v_pk_mad_u16", k_pk_mad_u16, 2, NACC * REPS},
    {"v_pk_mad_u16 op_sel", k_pk_mad_u16_opsel, 2, NACC * REPS},
    {"v_pk_mad_u16 sgpr-src", k_pk_mad_u16_sgpr, 2, NACC * REPS},
    {"v_pk_add_u16", k_pk_add_u16, 2, NACC * REPS},
    {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 2, NACC * REPS},
    {"v_mad_u32_u24", k_mad_u32_u24, 1, NACC * REPS},
    {"v_mad_u16", k_mad_u16, 1, NACC * REPS},
    {"v_add_u32", k_add_u32, 1, NACC * REPS},
    {"v_dot4_u32_u8", k_dot4_u32_u8, 4, NACC * REPS},
    {"v_dot4_i32_i8", k_dot4_i32_i8, 4, NACC * REPS},
    {"v_dot8_u32_u4", k_dot8_u32_u4, 8, NACC * REPS},
    {"v_dot2_u32_u16", k_dot2_u32_u16, 2, NACC * REPS},
    {"v_fma_f32", k_fma_f32, 1, NACC * REPS},
    {"v_pk_fma_f32", k_pk_fma_f32, 2, NACC * REPS},
  };
  const int iters = 2000;
  printf("%-24s %6s %12s %14s %12s\n", "instr", "w/SIMD", "ms", "Tinstr-lane/s", "TMAC/s");
  for (auto& c : cases) {
    for (int wps : {1, 2, 4, 8}) {
      int threads = 256;                  // 4 waves = one per SIMD
      int blocks = cus * wps;
      hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, in, out, 10);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      double lane_instr = (double)blocks * threads * iters * c.instr_per_iter;
      double rate = lane_instr / (best * 1e-3) / 1e12;
      printf("%-24s %6d %12.4f %14.3f %12.3f\n", c.name, wps, best, rate, rate * c.macs_per_lane_instr);
    }
  }
  printf("\nLDS ds_read_b32, per-lane dword stride K (bytes/clk/CU at 2.4 GHz nominal)\n");
  struct L { const char* name; kern_t k; } lds[] = {
    {"stride 1", k_lds_b32<1>}, {"stride 3", k_lds_b32<3>}, {"stride 4", k_lds_b32<4>}, {"stride 5", k_lds_b32<5>},
    {"stride 6", k_lds_b32<6>}, {"stride 7", k_lds_b32<7>}, {"stride 0 (bcast)", k_lds_b32<0>},
  };
  for (auto& c : lds) {
    for (int wps : {1, 2, 4}) {
      int threads = 256, blocks = cus * wps;
      hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, in, out, 10);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      double bytes = (double)blocks * threads * iters * 32 * 4;
      printf("%-24s %6d %12.4f %10.1f TB/s  %8.1f B/clk/CU\n", c.name, wps, best, bytes / (best * 1e-3) / 1e12,
             bytes / (best * 1e-3) / 2.4e9 / cus);
    }
  }
  return 0;
}
