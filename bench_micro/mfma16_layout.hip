#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
// checks the assumed operand / result layout of v_mfma_i32_16x16x64_i8: A lane l: row l&15, k = 16 (l>>4) .. +15 (bytes in order);
// B lane l: column l&15, same k; D lane l: column l&15, rows 4 (l>>4) + j in register j.
__global__ void k(const signed char *A, const signed char *B, int *D) {   // A[16][64], B[64][16] row-major, D[16][16]
  const int l = threadIdx.x, m = l & 15, kq = l >> 4;
  union { v4i v; signed char c[16]; } a, b;
  for (int j = 0; j < 16; j++) { a.c[j] = A[m * 64 + 16 * kq + j]; b.c[j] = B[(16 * kq + j) * 16 + m]; }
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, c, 0, 0, 0);
  for (int j = 0; j < 4; j++) D[(4 * kq + j) * 16 + m] = c[j];
}
int main() {
  signed char hA[16 * 64], hB[64 * 16]; int hD[256], ref[256];
  unsigned x = 5; for (auto &v : hA) { x = x * 1664525u + 1013904223u; v = (signed char)(x >> 24); } for (auto &v : hB) { x = x * 1664525u + 1013904223u; v = (signed char)(x >> 24); }
  for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) { int s = 0; for (int k = 0; k < 64; k++) s += hA[m * 64 + k] * hB[k * 16 + n]; ref[m * 16 + n] = s; }
  signed char *dA, *dB; int *dD; hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
  printf("layout of v_mfma_i32_16x16x64_i8 as assumed: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
  return 0;
}
