// Probe of direct global -> LDS loads on gfx950 (buffer_load_dwordx4 ... lds via __builtin_amdgcn_raw_ptr_buffer_load_lds):
//   (a) where the 64 x 16 bytes of one instruction land in LDS (M0 base + 16 * lane?), (b) what an out-of-range lane writes
//   (zeros?), (c) a source at a 16-byte aligned / dword-aligned / 2-byte aligned address, (d) whether vmcnt covers them.
// hipcc -O3 --offload-arch=gfx950 -o lds_dma lds_dma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int n, int rev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = 0xEE;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n, 0x00020000);
  for (int j = 0; j < 4; j++) {
    const int blk = wave * 4 + j;                          // 1 KB of LDS per instruction
    const int l = rev ? 63 - lane : lane;                  // rev: lane l reads the piece of lane 63 - l (does LDS placement follow the lane or the address?)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + blk * 1024), 16, blk * 1024 + 16 * l, 0, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);                           // vmcnt(0) lgkmcnt(0) expcnt(0)
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) dst[i] = lds[i];
}

int main() {
  const int n = 16 * 1024;
  std::vector<unsigned char> h(n + 64), out(n);
  for (int i = 0; i < n + 64; i++) h[i] = (unsigned char)(i * 7 + (i >> 8));
  unsigned char *d, *o; CK(hipMalloc(&d, n + 64)); CK(hipMalloc(&o, n));
  CK(hipMemcpy(d, h.data(), n + 64, hipMemcpyHostToDevice));
  for (int off : {0, 4, 2, 1, 3, 7}) for (int limit : {n, n - 1000}) for (int rev : {0, 1}) {
    CK(hipMemset(o, 0, n));
    k<<<1, 256, 16 * 1024>>>(d + off, o, limit, rev);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), o, n, hipMemcpyDeviceToHost));
    int same = 0, zero = 0, ee = 0, other = 0, first_bad = -1;
    for (int i = 0; i < n; i++) {
      int src_i = i;
      if (rev) { const int blk = i / 1024, in = i % 1024, l = in / 16; src_i = blk * 1024 + 16 * (63 - l) + in % 16; }
      const bool in_range = src_i < limit;
      if (in_range && out[i] == h[off + src_i]) same++;
      else if (!in_range && out[i] == 0) zero++;
      else if (out[i] == 0xEE) { ee++; if (first_bad < 0) first_bad = i; }
      else { other++; if (first_bad < 0) first_bad = i; }
    }
    printf("src offset %d, %5d bytes in range, lanes %s: as expected %5d, zero-filled past the range %5d, untouched %5d, other %5d (first unexpected byte %d)\n",
           off, limit, rev ? "reversed" : "in order", same, zero, ee, other, first_bad);
  }
  return 0;
}
