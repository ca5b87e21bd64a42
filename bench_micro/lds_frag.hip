// Micro-benchmark: LDS read cost of the operand-fragment access patterns considered for the int8 MFMA path.
//   P1  A fragment, padded rows:  lane (r = l&31, h = l>>5) reads 16 B at r*848 + 16h + 32*step   (16-byte aligned)
//   P2  A fragment, flat rows:    r*821 + 16h + 32*step                                           (unaligned)
//   P3  Toeplitz fragment from a reversed byte array: base - r + 16h + 32*step                    (unaligned, overlapping)
//   P4  as P3 but from 4 byte-shifted copies (pitch = 16 dwords mod 64) with 2 x ds_read2_b32     (4-byte aligned)
//   P5  ds_read_u8 of a flat byte image at (row(g,h)*821 + 32*step + r): the epilogue's m read
// hipcc -O3 --offload-arch=gfx950 -o lds_frag lds_frag.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int P>
__global__ void k_pat(int* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 40000 / 4; i += blockDim.x) ((unsigned*)lds)[i] = i * 2654435761u;
  __syncthreads();
  const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, w = threadIdx.x >> 6;
  unsigned base;
  if (P == 1) base = r * 848 + 16 * h;
  if (P == 2) base = r * 821 + 16 * h;
  if (P == 3) base = 64 - r + 16 * h;
  if (P == 4) { const unsigned o = 64 - r + 16 * h; base = (o & ~3u) + (o & 3u) * (4 * 464); }   // 464 dwords = 16 mod 64
  if (P == 5) base = h * 4 * 821 + r;
  base += (unsigned)(size_t)lds + w * 64;       // waves read slightly different places
  v4i acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 26; s++) {
      if (P <= 3) {
        v4i v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(32 * s));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc ^= v;
      } else if (P == 4) {
        unsigned long long v0, v1;
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v0) : "v"(base), "n"(8 * s), "n"(8 * s + 1));
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v1) : "v"(base), "n"(8 * s + 2), "n"(8 * s + 3));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc[0] ^= (int)v0; acc[1] ^= (int)(v0 >> 32); acc[2] ^= (int)v1; acc[3] ^= (int)(v1 >> 32);
      } else {
        unsigned v;
        asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(821 * (s & 3) + 32 * (s >> 2)));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        acc[0] ^= v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int* dout; CK(hipMalloc(&dout, (size_t)cus * 4 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name, int bytes_per_lane, int wgs_per_cu) {
    const int iters = 4000, blocks = cus * wgs_per_cu;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    kern<<<blocks, 256, 65536>>>(dout, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern<<<blocks, 256, 65536>>>(dout, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)wgs_per_cu * 4 * iters * 26;
    printf("%-44s WG/CU %d: %8.3f ms  %6.2f cycles per wave-read per CU @2.4GHz  %7.1f B/clk/CU\n", name, wgs_per_cu, ms,
           ms * 1e-3 * 2.4e9 / instr_per_cu, instr_per_cu * 64 * bytes_per_lane / (ms * 1e-3 * 2.4e9));
  };
  for (int wg : {1, 2}) {
    run(k_pat<1>, "P1 A frag padded (aligned b128)", 16, wg);
    run(k_pat<2>, "P2 A frag flat pitch 821 (unaligned b128)", 16, wg);
    run(k_pat<3>, "P3 Toeplitz frag reversed (unaligned b128)", 16, wg);
    run(k_pat<4>, "P4 Toeplitz frag 4 copies (2 x read2_b32)", 16, wg);
    run(k_pat<5>, "P5 epilogue byte read (ds_read_u8)", 1, wg);
  }
  return 0;
}
