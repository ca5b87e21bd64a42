// The probe that preceded k_verify_keys_m / k_product_tern_m / k_polymul_m (tools/peritem_mfma_model.py): ONE product with per-item
// operands on the int8 matrix cores, one item per wavefront, no shared key.
//   c = a * s in Z[x], a < q (two int8 digit planes), s ternary; rem = low + high, quot = -high (split by 1 - x^N).
// For a tile distance d = kb - ib one v_mfma_i32_32x32x32_i8 adds the contribution of every tile pair at that distance:
//   rows = output tile kb, columns = k', contraction = i';  A = chunk matrix of a digit plane shifted by d rows (aligned
//   16-byte reads from a zero-padded natural-order byte array), B = Toeplitz fragment of s (reversed cyclic array with 4
//   byte-shifted copies, as in kernel family 4).  The two digit planes go to separate accumulators (value = acc0 + 128 acc1),
//   so no operand needs scaling.  2 (2 NT - 1) + 2 matrix instructions per item.
// Measures items/s for the product alone (inputs in HBM, outputs written), checks a sample against a host convolution.
// hipcc -O3 --offload-arch=gfx950 -o peritem_mfma peritem_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned u32;

struct Geo { int N, NT, tpitch; };
#ifndef ABL
#define ABL 0   // timing-only variants: 1 no matrix loop, 2 reversed array built for the first item only, 4 no result stores, 8 no 3-period staging
#endif
constexpr int PAD_CH = 32;       // zero chunks on either side of the chunk matrix (tile distances reach +-(NT-1), rows 0..31)

static __host__ __device__ size_t fa_bytes(const Geo &g) { return (size_t)32 * (g.NT + 2 * PAD_CH); }
static __host__ __device__ size_t nat_bytes(const Geo &g) { return ((size_t)3 * g.N + 64 + 15) & ~(size_t)15; }
static __host__ __device__ size_t per_wave_bytes(const Geo &g) { return 2 * fa_bytes(g) + nat_bytes(g) + (size_t)16 * g.tpitch; }

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_num_vgpr(128))) void k_peritem(Geo g, u32 q, const uint16_t *__restrict__ a, const int8_t *__restrict__ s,
                                                        long B, uint16_t *__restrict__ rem, uint16_t *__restrict__ quot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  unsigned char *fa0 = lds + (size_t)wave * per_wave_bytes(g), *fa1 = fa0 + fa_bytes(g), *nat = fa1 + fa_bytes(g);
  u32 *T = (u32 *)(nat + nat_bytes(g));
  const int N = g.N, NT = g.NT, Y0 = 32 * NT - 1;
  for (size_t i = 16 * lane; i < 2 * fa_bytes(g); i += 16 * 64) *(v4i *)(fa0 + i) = (v4i){0, 0, 0, 0};   // pads stay zero
  const int hthr = (int)(q >> 1) - 65;
  const int y0 = Y0 - r + 16 * h;
  const u32 *tb = T + (y0 & 3) * g.tpitch + (y0 >> 2);
  u32 mlow[4];
  for (int c = 0; c < 4; c++) {
    u32 mk = 0;
    for (int jj = 0; jj < 4; jj++) mk |= (r >= 16 * h + 4 * c + jj) ? (0xFFu << (8 * jj)) : 0u;
    mlow[c] = mk;
  }
  for (long item = (long)blockIdx.x * WAVES + wave; item < B; item += (long)gridDim.x * WAVES) {
    const long row = item * N;
    // ---- stage: digit planes of a in natural order (A side), s three periods in natural order (source of the reversed array)
    if (16 * lane < 32 * NT) {
      union { v4i v; signed char c[16]; } d0, d1;
      union { v4i v; signed char c[16]; } sv;
      union { v4i v[2]; uint16_t c[16]; } av;
      // rows are only 2-byte / 1-byte aligned: unaligned 16-byte global loads work (slower than aligned ones; the
      // shipped kernels read aligned chunks and shift).  The last lanes read past the row: masked below, the buffers
      // of the probe are padded by one row.
      av.v[0] = *(const v4i *)(a + row + 16 * lane); av.v[1] = *(const v4i *)(a + row + 16 * lane + 8);
      sv.v = *(const v4i *)(s + row + 16 * lane);
      for (int j = 0; j < 16; j++) {
        const int i = 16 * lane + j;
        int hs = i < N ? (int)(av.c[j] & (q - 1)) : 0;
        hs = hs > hthr ? hs - (int)q : hs;
        const int lo = ((hs + 64) & 127) - 64;
        d0.c[j] = (signed char)lo;
        d1.c[j] = (signed char)((hs - lo) >> 7);
        sv.c[j] = i < N ? sv.c[j] : 0;
      }
      *(v4i *)(fa0 + 32 * PAD_CH + 16 * lane) = d0.v;
      *(v4i *)(fa1 + 32 * PAD_CH + 16 * lane) = d1.v;
      // three periods (+64 bytes) of s in natural order; period k starts at byte k N (any alignment): unaligned LDS stores
      if (!(ABL & 8) || item < (long)gridDim.x * WAVES)
      for (int k = 0; k < 3; k++)
        if (16 * lane + 16 <= N) *(v4i *)(nat + k * N + 16 * lane) = sv.v;
        else for (int j = 0; j < 16; j++) if (16 * lane + j < N) nat[k * N + 16 * lane + j] = (unsigned char)sv.c[j];
      if (lane < 4) *(v4i *)(nat + 3 * N + 16 * lane) = sv.v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- reversed cyclic array of s, 4 byte-shifted copies: T[c][w] = bytes rev[4w+c+j], rev[y] = s[(Y0 - y) mod N]
    if (!(ABL & 2) || item < (long)gridDim.x * WAVES)
    for (int x = lane; x < 4 * g.tpitch; x += 64) {
      const int c = x / g.tpitch, w = x - c * g.tpitch;
      int P = Y0 + 2 * N - (4 * w + c);
      P = P < 3 ? 3 : P;                                  // the pad words of a copy are never read by a fragment
      T[x] = __builtin_bswap32(*(const u32 *)(nat + P - 3));   // unaligned LDS read, bytes reversed
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- the product: one pass over the tile distances
    v16i L0, L1, H0, H1;
    for (int i = 0; i < 16; i++) { L0[i] = 0; L1[i] = 0; H0[i] = 0; H1[i] = 0; }
    const unsigned char *pa0 = fa0 + 32 * PAD_CH + 32 * r + 16 * h, *pa1 = fa1 + 32 * PAD_CH + 32 * r + 16 * h;
    if (!(ABL & 1)) {
    for (int d = -(NT - 1); d < 0; d++) {
      const u32 *p = tb - 8 * d;
      const v4i w = {(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
      H0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const v4i *)(pa0 - 32 * d), w, H0, 0, 0, 0);
      H1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const v4i *)(pa1 - 32 * d), w, H1, 0, 0, 0);
    }
    {
      const v4i w = {(int)tb[0], (int)tb[1], (int)tb[2], (int)tb[3]};
      const v4i wl = {(int)((u32)w[0] & mlow[0]), (int)((u32)w[1] & mlow[1]), (int)((u32)w[2] & mlow[2]), (int)((u32)w[3] & mlow[3])};
      const v4i wh = {(int)((u32)w[0] & ~mlow[0]), (int)((u32)w[1] & ~mlow[1]), (int)((u32)w[2] & ~mlow[2]), (int)((u32)w[3] & ~mlow[3])};
      const v4i a0 = *(const v4i *)pa0, a1 = *(const v4i *)pa1;
      L0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, wl, L0, 0, 0, 0);
      L1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, wl, L1, 0, 0, 0);
      H0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, wh, H0, 0, 0, 0);
      H1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, wh, H1, 0, 0, 0);
    }
    for (int d = 1; d < NT; d++) {
      const u32 *p = tb - 8 * d;
      const v4i w = {(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
      L0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const v4i *)(pa0 - 32 * d), w, L0, 0, 0, 0);
      L1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const v4i *)(pa1 - 32 * d), w, L1, 0, 0, 0);
    }
    }
    // ---- split by 1 - x^N and store: register i is output tile (i & 3) + 8 (i >> 2) + 4 h, column r
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int kb = (i & 3) + 8 * (i >> 2) + 4 * h, k = 32 * kb + r;
      if (kb < NT && k < N && (!(ABL & 4) || L0[i] == 0x7fffffff)) {
        const int lo = L0[i] + 128 * L1[i], hi = H0[i] + 128 * H1[i];
        rem[row + k] = (uint16_t)((u32)(lo + hi) & (q - 1));
        quot[row + k] = (uint16_t)((u32)(0 - hi) & (q - 1));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 821, q = argc > 2 ? atoi(argv[2]) : 4096;
  const long B = argc > 3 ? atol(argv[3]) : (1L << 18);
  Geo g; g.N = N; g.NT = (N + 31) / 32; g.tpitch = ((16 * g.NT + 31) / 32) * 32 + 8;
  if (N < 128 || N > 1024 || q > 8192) { printf("need 128 <= N <= 1024, q <= 8192\n"); return 1; }
  std::vector<uint16_t> a((size_t)(B + 1) * N); std::vector<int8_t> s((size_t)(B + 1) * N);
  srand(3);
  for (auto &x : a) x = (uint16_t)(rand() % q);
  for (auto &x : s) x = (int8_t)(rand() % 3 - 1);
  uint16_t *da, *drem, *dquot; int8_t *ds;
  CK(hipMalloc(&da, a.size() * 2)); CK(hipMalloc(&ds, s.size())); CK(hipMalloc(&drem, a.size() * 2)); CK(hipMalloc(&dquot, a.size() * 2));
  CK(hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, s.data(), s.size(), hipMemcpyHostToDevice));
  CK(hipMemset(drem, 0xEE, a.size() * 2)); CK(hipMemset(dquot, 0xEE, a.size() * 2));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  auto run = [&](auto kern, int WV) {
    const size_t lds = WV * per_wave_bytes(g);
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * WV, lds));
    const int blocks = prop.multiProcessorCount * (per_cu < 1 ? 1 : per_cu);
    kern<<<blocks, 64 * WV, lds>>>(g, (u32)q, da, ds, B, drem, dquot);
    CK(hipDeviceSynchronize());
    float b = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0));
      kern<<<blocks, 64 * WV, lds>>>(g, (u32)q, da, ds, B, drem, dquot);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      b = ms < b ? ms : b;
    }
    printf("N=%d q=%d B=%ld: %zu B of LDS per wave, %d workgroups of %d wave(s) per CU (%d waves per CU), grid %d: %.3f ms\n", N, q, B,
           per_wave_bytes(g), per_cu, WV, per_cu * WV, blocks, b);
    best = b < best ? b : best;
  };
  run(k_peritem<4>, 4);
  run(k_peritem<2>, 2);
  run(k_peritem<1>, 1);
  // check a sample against a direct convolution
  std::vector<uint16_t> hrem(a.size()), hquot(a.size());
  CK(hipMemcpy(hrem.data(), drem, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hquot.data(), dquot, a.size() * 2, hipMemcpyDeviceToHost));
  long bad = 0; int checked = 0;
  for (long it = 0; it < B; it += (B / 37 > 0 ? B / 37 : 1), checked++) {
    std::vector<long> lin(2 * N, 0);
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) lin[i + j] += (long)a[it * N + i] * s[it * N + j];
    for (int k = 0; k < N; k++) {
      const long rr = (((lin[k] + lin[N + k]) % q) + q) % q, qq = (((-lin[N + k]) % q) + q) % q;
      bad += hrem[it * N + k] != rr; bad += hquot[it * N + k] != qq;
    }
  }
  const double n_mfma = (double)B * (4.0 * (g.NT - 1) + 4.0);
  printf("per-item product on the matrix cores: %.3f ms for %ld items = %.1f M products/s; %.0f matrix instructions per item, %.0f G/s "
         "(%.1f %% of the 2.2 P MAC/s issue roof); %d items checked, %ld mismatches\n",
         best, B, B / best * 1e-3, n_mfma / B, n_mfma / best * 1e-6, n_mfma * 32768 / (best * 1e-3) / 2.2e15 * 100, checked, bad);
  printf("for comparison: k_polymul_split (packed-MAC vector ALU) runs the same product at 1.05 ms per 2^16 items = 62 M products/s\n");
  return bad != 0;
}
