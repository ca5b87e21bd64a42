// Micro-benchmark / layout probe for the int8 matrix path on gfx950:
//   (a) operand + result lane maps of v_mfma_i32_32x32x32_i8, checked with random data against a host product;
//   (b) issue rate of that instruction over the whole chip;
//   (c) what LDS reads return at addresses that are not naturally aligned (b32 / b64 / b128 / read2_b32);
//   (d) global dword / dwordx2 / dwordx4 stores at 2-byte-aligned addresses: correct? how fast?
// hipcc -O3 --offload-arch=gfx950 -o mfma_i8 mfma_i8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// hypothesis H: 0: k = 16*(l>>5) + j ; 1: k = 8*(l>>5) + (j&7) + 16*(j>>3)
__global__ void k_layout(const signed char* A, const signed char* B, int* D, int hyp) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  union { v4i v; signed char c[16]; } a, b;
  for (int j = 0; j < 16; j++) {
    const int k = hyp == 0 ? 16 * h + j : 8 * h + (j & 7) + 16 * (j >> 3);
    a.c[j] = A[r * 32 + k];          // A[row r][k]
    b.c[j] = B[k * 32 + r];          // B[k][col r]
  }
  v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, acc, 0, 0, 0);
  for (int g = 0; g < 16; g++) {
    const int row = (g & 3) + 8 * (g >> 2) + 4 * h, col = r;
    D[row * 32 + col] = acc[g];
  }
}

template <int NACC>
__global__ void k_rate(const int* in, int* out, int iters) {
  v4i a = {in[threadIdx.x & 63], in[1], in[2], in[3]}, b = {in[4], in[threadIdx.x & 31], in[6], in[7]};
  v16i acc[NACC];
  for (int t = 0; t < NACC; t++) for (int g = 0; g < 16; g++) acc[t][g] = t + g;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
  }
  int s = 0;
  for (int t = 0; t < NACC; t++) for (int g = 0; g < 16; g++) s ^= acc[t][g];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// (c) LDS reads at byte offset `off` from a 16-byte aligned base; image byte i holds (i & 255)
__global__ void k_lds_unaligned(unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char img[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) img[i] = (unsigned char)i;
  __syncthreads();
  if (threadIdx.x < 16) {
    const int off = threadIdx.x;
    unsigned addr = (unsigned)(size_t)(img + 64 + off);   // LDS byte address (low 32 bits of the shared pointer)
    unsigned r32, r64a, r64b, q0, q1, q2, q3, p0, p1;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r32) : "v"(addr));
    unsigned long long r64;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r64) : "v"(addr));
    r64a = (unsigned)r64; r64b = (unsigned)(r64 >> 32);
    v4i r128;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r128) : "v"(addr));
    q0 = r128[0]; q1 = r128[1]; q2 = r128[2]; q3 = r128[3];
    unsigned long long r2;
    asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r2) : "v"(addr));
    p0 = (unsigned)r2; p1 = (unsigned)(r2 >> 32);
    unsigned* o = out + 16 * off;
    o[0] = r32; o[1] = r64a; o[2] = r64b; o[3] = q0; o[4] = q1; o[5] = q2; o[6] = q3; o[7] = p0; o[8] = p1;
  }
}

// (d) stores of W dwords per lane at byte offset `mis` from natural alignment; region is contiguous
template <int W>
__global__ void k_store(unsigned char* base, int mis, long n_chunks, unsigned seed) {
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (long)gridDim.x * blockDim.x) {
    unsigned char* p = base + mis + c * (4 * W);
    const unsigned v = (unsigned)c * 2654435761u + seed;
    if (W == 1) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (W == 2) { unsigned long long vv = v | ((unsigned long long)(v + 1) << 32); asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(vv) : "memory"); }
    if (W == 4) { v4i vv = {(int)v, (int)(v + 1), (int)(v + 2), (int)(v + 3)}; asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(vv) : "memory"); }
  }
}
__global__ void k_store_short(unsigned short* base, long n, unsigned seed) {
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (long)gridDim.x * blockDim.x)
    base[c] = (unsigned short)(c * 40503u + seed);
}

int main() {
  // ---- (a)
  std::vector<signed char> A(1024), B(1024);
  srand(7);
  for (auto& x : A) x = (signed char)(rand() % 255 - 127);
  for (auto& x : B) x = (signed char)(rand() % 255 - 127);
  std::vector<int> ref(1024, 0), got(1024);
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int k = 0; k < 32; k++) s += (int)A[i * 32 + k] * (int)B[k * 32 + j]; ref[i * 32 + j] = s; }
  signed char *dA, *dB; int* dD;
  CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
  CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
  for (int hyp = 0; hyp < 2; hyp++) {
    k_layout<<<1, 64>>>(dA, dB, dD, hyp);
    CK(hipMemcpy(got.data(), dD, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; i++) bad += got[i] != ref[i];
    printf("layout hypothesis %d (%s): %d / 1024 mismatches\n", hyp, hyp == 0 ? "k = 16*(lane>>5) + j" : "k = 8*(lane>>5) + (j&7) + 16*(j>>3)", bad);
  }
  // ---- (b)
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int *din, *dout; CK(hipMalloc(&din, 1024)); CK(hipMemset(din, 1, 1024)); CK(hipMalloc(&dout, (size_t)cus * 8 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto rate = [&](auto kern, int nacc, int waves_per_simd, const char* name) {
    const int iters = 20000, blocks = cus * waves_per_simd;
    kern<<<blocks, 256>>>(din, dout, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern<<<blocks, 256>>>(din, dout, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)blocks * 4 * iters * nacc;
    printf("%-28s waves/SIMD %d: %8.3f ms  %7.2f G mfma/s  %8.1f T MAC/s  (%.1f cycles/mfma/SIMD @2.4GHz)\n", name, waves_per_simd, ms, n / ms * 1e-6,
           n * 32768 / ms * 1e-9, (double)ms * 1e-3 * 2.4e9 / ((double)iters * nacc * waves_per_simd));
  };
  rate(k_rate<1>, 1, 1, "mfma_i32_32x32x32_i8 x1 acc");
  rate(k_rate<4>, 4, 1, "mfma_i32_32x32x32_i8 x4 acc");
  rate(k_rate<4>, 4, 2, "mfma_i32_32x32x32_i8 x4 acc");
  rate(k_rate<8>, 8, 1, "mfma_i32_32x32x32_i8 x8 acc");
  // ---- (c)
  unsigned* dl; CK(hipMalloc(&dl, 16 * 16 * 4));
  k_lds_unaligned<<<1, 64>>>(dl);
  std::vector<unsigned> hl(256); CK(hipMemcpy(hl.data(), dl, 1024, hipMemcpyDeviceToHost));
  printf("LDS reads at base+64+off (image byte i = i&255); expected first byte = 64+off\n");
  for (int off = 0; off < 16; off++) {
    const unsigned* o = &hl[16 * off];
    printf(" off %2d: b32 %08x | b64 %08x %08x | b128 %08x %08x %08x %08x | read2_b32 %08x %08x\n", off, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]);
  }
  // ---- (d)
  const long bytes = 1L << 30;
  unsigned char* big; CK(hipMalloc(&big, bytes + 64));
  auto st = [&](int W, int mis) {
    const long n = bytes / (4 * W);
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      if (W == 1) k_store<1><<<cus * 8, 256>>>(big, mis, n, 5);
      if (W == 2) k_store<2><<<cus * 8, 256>>>(big, mis, n, 5);
      if (W == 4) k_store<4><<<cus * 8, 256>>>(big, mis, n, 5);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // verify a sample
    std::vector<unsigned char> h(4096 + 64); CK(hipMemcpy(h.data(), big, h.size(), hipMemcpyDeviceToHost));
    int bad = 0;
    for (long c = 0; c < 4096 / (4 * W); c++) for (int w = 0; w < W; w++) {
      unsigned v; memcpy(&v, &h[mis + c * 4 * W + 4 * w], 4);
      bad += v != (unsigned)c * 2654435761u + 5 + w;
    }
    printf("store dwordx%d at +%d bytes: %7.3f ms  %7.1f GB/s  sample mismatches %d\n", W, mis, ms, bytes / ms * 1e-6, bad);
  };
  for (int W : {1, 2, 4}) for (int mis : {0, 2, 1}) st(W, mis);
  {
    for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); k_store_short<<<cus * 8, 256>>>((unsigned short*)big, bytes / 2, 3); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("store short (coalesced)      : %7.3f ms  %7.1f GB/s\n", ms, bytes / ms * 1e-6);
  }
  return 0;
}
