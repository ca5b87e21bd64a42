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

extern "C" int clock_probe_read(void *stream, unsigned long long *d_out) {
  hipLaunchKernelGGL(k_read_clocks, dim3(1), dim3(1), 0, (hipStream_t)stream, d_out);
  return (int)hipGetLastError();
}
// mode 0: one idle wave; 1: `blocks` workgroups of `threads` lanes issuing matrix instructions back to back
extern "C" int clock_probe_spin(void *stream, int mode, int blocks, int threads, unsigned long long ticks, unsigned long long *d_out, const int *d_in) {
  if (mode == 0) hipLaunchKernelGGL(k_spin<0>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks, d_out, d_in);
  else hipLaunchKernelGGL(k_spin<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks, d_out, d_in);
  return (int)hipGetLastError();
}
