// Clock probe for the "power limit or latency?" question (missing #4).
//   s_memtime     = the shader-clock counter the phase stamps use (tools/phase_stamps.py)
//   s_memrealtime = the constant 100 MHz reference counter
// A one-lane kernel reads both; launched on the engine's stream before and after a run of product kernels, the two deltas
// give the AVERAGE shader clock over that run: d(memtime) / d(memrealtime) x 100 MHz.  k_spin_* hold the chip in a known state
// for a fixed real time (one wave: an idle chip at its top clock; every SIMD issuing v_mfma_i32_32x32x32_i8 back to back: the
// matrix pipes at full rate, nothing else) and report their own pair of deltas per workgroup.
// hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libclock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k_read_clocks(unsigned long long *out) {
  out[0] = __builtin_amdgcn_s_memtime();
  out[1] = __builtin_amdgcn_s_memrealtime();
}

// Spins until `ticks` of the 100 MHz counter have passed; out[block] = {d memtime, d memrealtime, iterations}.
template <int MFMA>
__global__ void k_spin(unsigned long long ticks, unsigned long long *out, const int *in) {
  v4i a = {in[threadIdx.x & 63], in[1], in[2], in[3]}, b = {in[4], in[threadIdx.x & 31], in[6], in[7]};
  v16i acc[4];
  for (int t = 0; t < 4; t++) for (int g = 0; g < 16; g++) acc[t][g] = t + g;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long it = 0, r1 = r0;
  while (r1 - r0 < ticks) {
    if (MFMA) {
      for (int k = 0; k < 64; k++) {
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
      }
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
    it++;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int t = 0; t < 4; t++) for (int g = 0; g < 16; g++) s ^= acc[t][g];
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x + 0] = t1 - t0;
    out[3 * blockIdx.x + 1] = r1 - r0;
    out[3 * blockIdx.x + 2] = it + (s == 0x7fffffff);
  }
}

// The same spin with v_mfma_scale_f32_32x32x64_f8f6f4 on fp4 (e2m1) operands: K = 64 per instruction, fp32 accumulation (exact for
// ternary x ternary sums), scales 2^0.  out as k_spin.
__global__ void k_spin_fp4(unsigned long long ticks, unsigned long long *out, const int *in) {
  v8i a = {in[threadIdx.x & 63], in[1], in[2], in[3], 0, 0, 0, 0}, b = {in[4], in[threadIdx.x & 31], in[6], in[7], 0, 0, 0, 0};
  v16f acc[4];
  for (int t = 0; t < 4; t++) for (int g = 0; g < 16; g++) acc[t][g] = (float)(t + g);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long it = 0, r1 = r0;
  while (r1 - r0 < ticks) {
    for (int k = 0; k < 64; k++) {
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 4, 4, 0, 127, 0, 127);
    }
    it++;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < 4; t++) for (int g = 0; g < 16; g++) s += acc[t][g];
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x + 0] = t1 - t0;
    out[3 * blockIdx.x + 1] = r1 - r0;
    out[3 * blockIdx.x + 2] = it + (s == 12345.f);
  }
}

// Energy per instruction class: the same timed spin with one kind of instruction issued back to back by every wave.
//   KIND 3: v_perm_b32 (VOP3, 4 cycles per wave)   4: v_add_u32 (VOP2)   5: ds_read_b128 (aligned, conflict-free)
//   KIND 6: ds_read_u8 at scattered addresses (table lookups)   7: ds_write_b16   8: ds_write_b64   9: s_barrier-free idle issue (s_nop)
// out[block] = {d memtime, d memrealtime, loop iterations}; one iteration = 256 instructions of the kind per wave.
template <int KIND>
__global__ void k_spin_kind(unsigned long long ticks, unsigned long long *out, const int *in) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((int *)sm)[i] = in[i & 63];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned x0 = in[lane], x1 = in[(lane + 1) & 63], x2 = in[(lane + 2) & 63], x3 = in[(lane + 3) & 63];
  const unsigned a128 = (unsigned)(size_t)(sm + ((wave * 1024 + lane * 16) & 32767));
  unsigned a8 = (unsigned)(size_t)(sm + ((lane * 517 + wave * 61) & 8191));
  const unsigned a16 = (unsigned)(size_t)(sm + ((wave * 2048 + lane * 2) & 32767)), a64 = (unsigned)(size_t)(sm + ((wave * 2048 + lane * 8) & 32767));
  typedef int v4 __attribute__((ext_vector_type(4)));
  v4 q = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long it = 0, r1 = r0;
  while (r1 - r0 < ticks) {
    for (int k = 0; k < 64; k++) {
      if (KIND == 3) {
        asm volatile("v_perm_b32 %0, %0, %1, %2\n\tv_perm_b32 %1, %1, %2, %3\n\tv_perm_b32 %2, %2, %3, %0\n\tv_perm_b32 %3, %3, %0, %1"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      } else if (KIND == 4) {
        asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      } else if (KIND == 5) {
        v4 t0_, t1_, t2_, t3_;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(t0_), "=v"(t1_), "=v"(t2_), "=v"(t3_) : "v"(a128) : "memory");
        q ^= t0_ ^ t1_ ^ t2_ ^ t3_;
      } else if (KIND == 6) {
        unsigned b0, b1, b2, b3;
        asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:4099\n\tds_read_u8 %2, %4 offset:8205\n\tds_read_u8 %3, %4 offset:12311\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(a8) : "memory");
        x0 ^= b0 ^ b1 ^ b2 ^ b3;
      } else if (KIND == 7) {
        asm volatile("ds_write_b16 %0, %1\n\tds_write_b16 %0, %1 offset:128\n\tds_write_b16 %0, %1 offset:256\n\tds_write_b16 %0, %1 offset:384\n\ts_waitcnt lgkmcnt(0)"
                     :: "v"(a16), "v"(x0) : "memory");
      } else if (KIND == 8) {
        unsigned long long v = ((unsigned long long)x1 << 32) | x0;
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:512\n\tds_write_b64 %0, %1 offset:1024\n\tds_write_b64 %0, %1 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                     :: "v"(a64), "v"(v) : "memory");
      } else {
        asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
      }
    }
    it++;
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x + 0] = t1 - t0;
    out[3 * blockIdx.x + 1] = r1 - r0;
    out[3 * blockIdx.x + 2] = it + ((x0 ^ x1 ^ x2 ^ x3 ^ (unsigned)q[0] ^ (unsigned)q[1] ^ (unsigned)q[2] ^ (unsigned)q[3]) == 0x12345u);
  }
}

extern "C" int clock_probe_read(void *stream, unsigned long long *d_out) {
  hipLaunchKernelGGL(k_read_clocks, dim3(1), dim3(1), 0, (hipStream_t)stream, d_out);
  return (int)hipGetLastError();
}
// mode 0: one idle wave; 1: `blocks` workgroups of `threads` lanes issuing matrix instructions back to back
extern "C" int clock_probe_spin(void *stream, int mode, int blocks, int threads, unsigned long long ticks, unsigned long long *d_out, const int *d_in) {
#define KIND_CASE(K) if (mode == K) { hipLaunchKernelGGL(k_spin_kind<K>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks, d_out, d_in); return (int)hipGetLastError(); }
  KIND_CASE(3) KIND_CASE(4) KIND_CASE(5) KIND_CASE(6) KIND_CASE(7) KIND_CASE(8) KIND_CASE(9)
  if (mode == 2) hipLaunchKernelGGL(k_spin_fp4, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks, d_out, d_in);
  else if (mode == 0) hipLaunchKernelGGL(k_spin<0>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks, d_out, d_in);
  else hipLaunchKernelGGL(k_spin<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks, d_out, d_in);
  return (int)hipGetLastError();
}
