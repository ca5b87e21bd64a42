// ntru_engine.hip -- MI355X (gfx950) kernels and C ABI of the NTRU polynomial-ring engine.
//
// Hot path of numtel/ntru-circom re-designed for CDNA4 (reference: index.js):
//   multiplyPolynomials (index.js:319-355)  -> exact integer schoolbook in Z/2^16 (q | 2^16, so natural u16
//                                              wrap-around IS the mod-q residue; mod-p sums stay < 2^16)
//   dividePolynomials by I=1-x^N (:358-401) -> closed form fused into the product's epilogue
//   addPolynomials (:235-244), lift (:117)  -> fused into the same epilogue
//
// Work decomposition ("one ciphertext per wavefront"):
//   * an item (one polynomial product) is owned by nl = ceil(N / 2K) consecutive lanes of a wave; lane s owns
//     the 2K outputs k in [2K s, 2K s + 2K) as K packed u16 pairs (v_pk_mad_u16: 2 MACs per lane-instruction).
//     For small N several items share a wave (G = 64 / nl).
//   * the "window" operand b lives in LDS as EO[u] = { E[u] = (bc[2u], bc[2u+1]), O[u] = (bc[2u-1], bc[2u]) },
//     bc = b extended cyclically with period N, so that for step i the K pairs a lane needs are K consecutive
//     8-byte entries, and going from step i to i+2 slides that window by exactly one entry: one ds_read_b64 per
//     lane per two steps, lane stride K entries (K odd => conflict-free, profiles/r01_microbench_valu_lds.txt).
//   * the "broadcast" operand a is read from LDS two coefficients at a time and applied with op_sel splats.
//   * T[k] = sum_i a[i] bc[k-i] is the cyclic product = remainder; the low half c[k] of the LINEAR product (needed
//     for the quotient) is T's value just before the lane's own block of i plus an in-block
//     triangle; high = T - low; quotient = -high.  (tools/lane_model.py is the executable spec of this indexing.)
//
// No CPU fallback exists in this file: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>

#include "engine_internal.h"

typedef unsigned short u16;
typedef unsigned int u32;
typedef u16 u16x2 __attribute__((ext_vector_type(2)));

#define WAVES_PER_BLOCK 4
#define BLOCK_THREADS (WAVES_PER_BLOCK * 64)

struct Geom {
  int N;       // ring size
  int nl;      // lanes per item = ceil(N / 2K)
  int G;       // items per wave = 64 / nl
  int off;     // K*nl: position of logical entry u = 0 inside an EO array
  int eo_len;  // 2*K*nl 8-byte entries per EO array
  int a_len;   // K*nl dwords (2K*nl u16) per staged a-operand
};

static __device__ __forceinline__ u16x2 as_pair(u32 v) { return __builtin_bit_cast(u16x2, v); }
static __device__ __forceinline__ u32 as_u32(u16x2 v) { return __builtin_bit_cast(u32, v); }

// x mod a small runtime modulus; p = 3 (every NTRU parameter set) gets the constant-divisor sequence.
static __device__ __forceinline__ u32 mod_small(u32 x, u32 m) { return m == 3u ? x % 3u : x % m; }

// Order this wave's LDS writes before its later LDS reads (regions touched here are private to one wave).
static __device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// ---- operand value functors: coefficient j of an operand, already mapped into its u16 representation ----------
struct ValU16 {            // e, h, fq, generic a/b
  const u16 *p;
  __device__ __forceinline__ u32 operator()(int j) const { return p[j]; }
};
struct ValU16x3 {          // fqp = p*fq left unreduced (index.js:155); 3*8191 < 2^16
  const u16 *p; u32 mul;
  __device__ __forceinline__ u32 operator()(int j) const { return (u32)p[j] * mul; }
};
struct ValU16x3m {         // (p*fq) mod q: the same window reduced, for the add path's 16-bit field budget
  const u16 *p; u32 mul, mask;
  __device__ __forceinline__ u32 operator()(int j) const { return ((u32)p[j] * mul) & mask; }
};
struct ValU8 {             // r, fp, m
  const uint8_t *p;
  __device__ __forceinline__ u32 operator()(int j) const { return p[j]; }
};
struct ValTernary {        // f, g in {-1,0,1}: -1 -> mod-1 (index.js:112,151,152,156)
  const int8_t *p; u32 neg;
  __device__ __forceinline__ u32 operator()(int j) const { int v = p[j]; return v < 0 ? neg : (u32)v; }
};

struct ValLds {            // an operand staged as u16 in LDS (per-item windows: coalesced global loads first)
  const u16 *p;
  __device__ __forceinline__ u32 operator()(int j) const { return p[j]; }
};

// Stage N coefficients of a per-item operand into LDS as u16 (lane `sub` of the item's `nl` lanes; coalesced).
template <class F>
static __device__ __forceinline__ void stage_raw(u16 *raw, int N, int nl, F val, int sub, bool active) {
  if (!active) return;
  for (int i = sub; i < N; i += nl) raw[i] = (u16)val(i);
}

// Build one EO array (see header) from `val`; executed by `nthr` cooperating threads, this one being `tid`.
template <class F>
static __device__ __forceinline__ void build_eo(uint2 *eo, const Geom &g, F val, int tid, int nthr) {
  const int N = g.N;
  for (int x = tid; x < g.eo_len; x += nthr) {
    int j = 2 * (x - g.off);                 // |j| < 2*off <= N + 2K <= 2N
    j += j < 0 ? N : 0; j += j < 0 ? N : 0;
    j -= j >= N ? N : 0; j -= j >= N ? N : 0;
    int jm = j == 0 ? N - 1 : j - 1;
    int jp = j + 1 == N ? 0 : j + 1;
    u32 c0 = val(j) & 0xFFFFu, cm = val(jm) & 0xFFFFu, cp = val(jp) & 0xFFFFu;
    eo[x] = make_uint2(c0 | (cp << 16), cm | (c0 << 16));
  }
}

// Stage the broadcast operand as zero-padded u16s: a16[i] = i < N ? val(i) : 0 for i < 2*a_len.
template <class F>
static __device__ __forceinline__ void stage_a(u16 *a16, const Geom &g, F val, int sub, bool active) {
  if (!active) return;
  const int n2 = 2 * g.a_len;
  for (int i = sub; i < n2; i += g.nl) a16[i] = i < g.N ? (u16)val(i) : (u16)0;
}

// The O(N^2) accumulate.  eo: this item's EO array, a32: its staged a-operand (packed pairs), sub: lane's index in item.
template <int K>
static __device__ __forceinline__ void mac_core(const uint2 *__restrict__ eo, const u32 *__restrict__ a32,
                                                const Geom &g, int sub, bool want_low,
                                                u16x2 (&T)[K], u16x2 (&low)[K]) {
  u32 WE[K], WO[K];
  const uint2 *nb = eo + (K * sub + g.off);     // logical entry K*sub - K*m, m = 0
#pragma unroll
  for (int x = 0; x < K; x++) { uint2 v = nb[x]; WE[x] = v.x; WO[x] = v.y; }
#pragma unroll
  for (int t = 0; t < K; t++) { T[t] = (u16x2){0, 0}; low[t] = (u16x2){0, 0}; }
  const int nblk = g.nl;
  for (int m = 0; m < nblk; m++) {
    if (want_low && m == sub) {                 // snapshot: everything accumulated so far has i < 2K*sub <= k
#pragma unroll
      for (int t = 0; t < K; t++) low[t] = T[t];
    }
    uint2 nw[K]; u32 av[K];
#pragma unroll
    for (int s = 0; s < K; s++) nw[s] = nb[-1 - s];
#pragma unroll
    for (int s = 0; s < K; s++) av[s] = a32[K * m + s];
#pragma unroll
    for (int s = 0; s < K; s++) {
      const u16x2 ap = as_pair(av[s]);
#pragma unroll
      for (int t = 0; t < K; t++) T[t] = ap.xx * as_pair(WE[(t - s + K) % K]) + T[t];   // i = 2(Km+s)
#pragma unroll
      for (int t = 0; t < K; t++) T[t] = ap.yy * as_pair(WO[(t - s + K) % K]) + T[t];   // i + 1
      WE[K - 1 - s] = nw[s].x; WO[K - 1 - s] = nw[s].y;
    }
    nb -= K;
  }
}

// In-block triangle: d[k0+j] = sum_{u<=j} a[k0+u] * b[j-u] for the lane's own 2K outputs (k0 = 2K*sub).
template <int K>
static __device__ __forceinline__ void diag_core(const uint2 *__restrict__ eo, const u32 *__restrict__ a32,
                                                 const Geom &g, int sub, u16x2 (&d)[K]) {
  u32 ZE[K], ZO[K];
#pragma unroll
  for (int x = 0; x < K; x++) { uint2 v = eo[g.off + x]; ZE[x] = v.x; ZO[x] = v.y; }
  ZO[0] &= 0xFFFF0000u;                          // O[0] = (b[-1], b[0]): b[-1] does not exist in the linear product
#pragma unroll
  for (int t = 0; t < K; t++) d[t] = (u16x2){0, 0};
#pragma unroll
  for (int s = 0; s < K; s++) {
    const u16x2 ap = as_pair(a32[K * sub + s]);
#pragma unroll
    for (int t = s; t < K; t++) {
      d[t] = ap.xx * as_pair(ZE[t - s]) + d[t];
      d[t] = ap.yy * as_pair(ZO[t - s]) + d[t];
    }
  }
}

// One product a*b with split by 1-x^N, results left in registers as K pairs per lane.
//   rem  = (T + addend) mod `mod`      quot = (-high) mod `mod`
// POW2: mod is a power of two (mask arithmetic on the wrapped u16 sums); otherwise sums are exact and `mod` small.
template <int K, bool POW2>
static __device__ __forceinline__ void product_split(const uint2 *eo, const u32 *a32, const Geom &g, int sub,
                                                     bool want_quot, u32 mod, u16x2 (&rem)[K], u16x2 (&quot)[K]) {
  u16x2 T[K], low[K];
  mac_core<K>(eo, a32, g, sub, want_quot, T, low);
  if (want_quot) {
    u16x2 d[K];
    diag_core<K>(eo, a32, g, sub, d);
#pragma unroll
    for (int t = 0; t < K; t++) {
      u16x2 hi = T[t] - (low[t] + d[t]);
      if (POW2) {
        quot[t] = ((u16x2){0, 0} - hi) & (u16)(mod - 1);
      } else {
        u32 h0 = mod_small(hi.x, mod), h1 = mod_small(hi.y, mod);
        quot[t] = (u16x2){(u16)(h0 ? mod - h0 : 0), (u16)(h1 ? mod - h1 : 0)};
      }
    }
  }
#pragma unroll
  for (int t = 0; t < K; t++) {
    if (POW2) rem[t] = T[t];                       // masked by the caller after the optional addend
    else rem[t] = (u16x2){(u16)mod_small(T[t].x, mod), (u16)mod_small(T[t].y, mod)};
  }
}

template <int K, class OutT>
static __device__ __forceinline__ void store_pairs(OutT *row, const Geom &g, int sub, const u16x2 (&v)[K]) {
#pragma unroll
  for (int t = 0; t < K; t++) {
    int k = 2 * K * sub + 2 * t;
    if (k < g.N) row[k] = (OutT)v[t].x;
    if (k + 1 < g.N) row[k + 1] = (OutT)v[t].y;
  }
}

struct LaneId {
  int wave, lane, grp, sub; bool active;
};
static __device__ __forceinline__ LaneId lane_id(const Geom &g) {
  LaneId L;
  L.wave = threadIdx.x >> 6;     // (readfirstlane here saves 5 VGPRs in k_encrypt_t but measured 11% slower)
  L.lane = threadIdx.x & 63;
  L.active = L.lane < g.G * g.nl;
  L.grp = L.active ? L.lane / g.nl : 0;
  L.sub = L.active ? L.lane - L.grp * g.nl : 0;
  return L;
}

// ---- kernels ------------------------------------------------------------------------------------------------
// dynamic LDS: [shared EO arrays][per-wave regions]; per-wave = G a-operands (+ G EO arrays for per-item windows)

// encryptBits, index.js:87-110: e = (m + r*h) mod q split by I.
template <int K>
__global__ __launch_bounds__(BLOCK_THREADS) void k_encrypt(Geom g, u32 q, const u16 *__restrict__ h,
                                                           const uint8_t *__restrict__ r,
                                                           const uint8_t *__restrict__ m, long B,
                                                           u16 *__restrict__ e, u16 *__restrict__ quotE) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint2 *eo_h = (uint2 *)lds;
  const LaneId L = lane_id(g);
  u32 *a_wave = (u32 *)(lds + (size_t)g.eo_len * 8) + (size_t)L.wave * g.G * g.a_len;
  u32 *a32 = a_wave + (size_t)L.grp * g.a_len;
  build_eo(eo_h, g, ValU16{h}, threadIdx.x, BLOCK_THREADS);
  __syncthreads();
  const long ngroups = (B + g.G - 1) / g.G;
  const bool want_quot = quotE != nullptr;
  for (long grp = (long)blockIdx.x * WAVES_PER_BLOCK + L.wave; grp < ngroups; grp += (long)gridDim.x * WAVES_PER_BLOCK) {
    const long item = grp * g.G + L.grp;
    const bool valid = L.active && item < B;
    const long row = (valid ? item : 0) * g.N;
    stage_a((u16 *)a32, g, ValU8{r + row}, L.sub, L.active);
    wave_lds_fence();
    u16x2 rem[K], quot[K];
    product_split<K, true>(eo_h, a32, g, L.sub, want_quot, q, rem, quot);
#pragma unroll
    for (int t = 0; t < K; t++) {
      int k = 2 * K * L.sub + 2 * t;
      u16x2 add = {(u16)(k < g.N ? m[row + k] : 0), (u16)(k + 1 < g.N ? m[row + k + 1] : 0)};
      rem[t] = (rem[t] + add) & (u16)(q - 1);
    }
    if (valid) {
      store_pairs<K>(e + row, g, L.sub, rem);
      if (want_quot) store_pairs<K>(quotE + row, g, L.sub, quot);
    }
    wave_lds_fence();
  }
}

// decryptBits, index.js:111-140: a = f*e mod q; split; lift; c = fp*b mod p; split.
template <int K>
__global__ __launch_bounds__(BLOCK_THREADS) void k_decrypt(Geom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                           const uint8_t *__restrict__ fp,
                                                           const u16 *__restrict__ e, long B,
                                                           uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                           u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint2 *eo_f = (uint2 *)lds;
  uint2 *eo_fp = eo_f + g.eo_len;
  const LaneId L = lane_id(g);
  u32 *a_wave = (u32 *)(lds + (size_t)g.eo_len * 16) + (size_t)L.wave * g.G * g.a_len;
  u32 *a32 = a_wave + (size_t)L.grp * g.a_len;
  build_eo(eo_f, g, ValTernary{f, q - 1}, threadIdx.x, BLOCK_THREADS);
  build_eo(eo_fp, g, ValU8{fp}, threadIdx.x, BLOCK_THREADS);
  __syncthreads();
  const long ngroups = (B + g.G - 1) / g.G;
  const bool want_q1 = quot1 != nullptr, want_q2 = quot2 != nullptr;
  for (long grp = (long)blockIdx.x * WAVES_PER_BLOCK + L.wave; grp < ngroups; grp += (long)gridDim.x * WAVES_PER_BLOCK) {
    const long item = grp * g.G + L.grp;
    const bool valid = L.active && item < B;
    const long row = (valid ? item : 0) * g.N;
    stage_a((u16 *)a32, g, ValU16{e + row}, L.sub, L.active);
    wave_lds_fence();
    u16x2 r1[K], q1[K];
    product_split<K, true>(eo_f, a32, g, L.sub, want_q1, q, r1, q1);
#pragma unroll
    for (int t = 0; t < K; t++) r1[t] = r1[t] & (u16)(q - 1);
    if (valid) {
      if (rem1) store_pairs<K>(rem1 + row, g, L.sub, r1);
      if (want_q1) store_pairs<K>(quot1 + row, g, L.sub, q1);
    }
    // centred lift, index.js:117 verbatim: x > q/2 ? (x+1)%p : x%p ; zero beyond N so the padding stays zero
    wave_lds_fence();
    if (L.active) {
#pragma unroll
      for (int t = 0; t < K; t++) {
        int k = 2 * K * L.sub + 2 * t;
        u32 x0 = r1[t].x, x1 = r1[t].y;
        u32 b0 = mod_small(2 * x0 > q ? x0 + 1 : x0, p), b1 = mod_small(2 * x1 > q ? x1 + 1 : x1, p);
        b0 = k < g.N ? b0 : 0; b1 = k + 1 < g.N ? b1 : 0;
        a32[K * L.sub + t] = b0 | (b1 << 16);
      }
    }
    wave_lds_fence();
    u16x2 r2[K], q2[K];
    product_split<K, false>(eo_fp, a32, g, L.sub, want_q2, p, r2, q2);
    if (valid) {
      store_pairs<K>(value + row, g, L.sub, r2);
      if (want_q2) store_pairs<K>(quot2 + row, g, L.sub, q2);
    }
    wave_lds_fence();
  }
}

// generic a*b mod `mod` split by I with per-item operands (multiplyPolynomials + dividePolynomials(.,I,.)).
// PUBKEY: generatePublicKeyH (index.js:72-79) for per-item keys: a = g in {-1,0,1} (int8, passed through `a`),
// b = p*fq reduced mod q (scale = p), only the remainder (= h before trimming) is stored.
template <int K, bool PUBKEY = false>
__global__ __launch_bounds__(BLOCK_THREADS) void k_polymul_split(Geom g, u32 mod, int pow2,
                                                                 const u16 *__restrict__ a, const u16 *__restrict__ b,
                                                                 long B, u16 *__restrict__ quot, u16 *__restrict__ rem,
                                                                 u32 scale = 1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const LaneId L = lane_id(g);
  const size_t raw_len = ((size_t)g.N + 1) & ~(size_t)1;                 // u16 slots per staged operand (dword aligned)
  const size_t per_wave = (size_t)g.G * ((size_t)g.eo_len * 8 + (size_t)g.a_len * 4 + raw_len * 2);
  unsigned char *wbase = lds + (size_t)L.wave * per_wave;
  uint2 *eo = (uint2 *)wbase + (size_t)L.grp * g.eo_len;
  u32 *a32 = (u32 *)(wbase + (size_t)g.G * g.eo_len * 8) + (size_t)L.grp * g.a_len;
  u16 *raw = (u16 *)(wbase + (size_t)g.G * ((size_t)g.eo_len * 8 + (size_t)g.a_len * 4)) + (size_t)L.grp * raw_len;
  const long ngroups = (B + g.G - 1) / g.G;
  for (long grp = (long)blockIdx.x * WAVES_PER_BLOCK + L.wave; grp < ngroups; grp += (long)gridDim.x * WAVES_PER_BLOCK) {
    const long item = grp * g.G + L.grp;
    const bool valid = L.active && item < B;
    const long row = (valid ? item : 0) * g.N;
    if (PUBKEY) {
      stage_raw(raw, g.N, g.nl, ValU16x3m{b + row, scale, mod - 1}, L.sub, L.active);
      stage_a((u16 *)a32, g, ValTernary{(const int8_t *)a + row, mod - 1}, L.sub, L.active);
    } else {
      stage_raw(raw, g.N, g.nl, ValU16{b + row}, L.sub, L.active);
      stage_a((u16 *)a32, g, ValU16{a + row}, L.sub, L.active);
    }
    wave_lds_fence();
    if (L.active) build_eo(eo, g, ValLds{raw}, L.sub, g.nl);
    wave_lds_fence();
    u16x2 r[K], qv[K];
    if (pow2) {
      product_split<K, true>(eo, a32, g, L.sub, !PUBKEY, mod, r, qv);
#pragma unroll
      for (int t = 0; t < K; t++) r[t] = r[t] & (u16)(mod - 1);
    } else {
      product_split<K, false>(eo, a32, g, L.sub, true, mod, r, qv);
    }
    if (valid) {
      store_pairs<K>(rem + row, g, L.sub, r);
      if (!PUBKEY) store_pairs<K>(quot + row, g, L.sub, qv);
    }
    wave_lds_fence();
  }
}


// ================================================================================================================
// Ternary-stepping kernels ("add path").
//
// Every product on the hot path has one TERNARY operand (r, f, g, the lifted message), and on gfx950 every packed /
// multiply VALU op issues at 4 cycles per wave while a plain v_add_u32 issues at 2 (profiles/r01_microbench_valu_lds.txt).
// So the ternary operand becomes the stepping operand: it is turned into 2-bit codes (0 skip, 1 "+w into S1",
// 2 "+w into S2"; the other symbol c = 2 or -1 is applied once at the end, T = S1 + c*S2), the codes of one block of
// 2K steps live in one wave-uniform dword, zero steps are skipped by a scalar branch, and the windowed operand is
// accumulated with v_add_u32 on two packed 16-bit fields.  A field may only hold `limit` additions of values < q on
// top of a masked value before it could carry into its neighbour; popcounts of the code word keep that budget and
// the accumulators are masked (mod q is free: q | 2^16) only when the next K steps could exceed it.
// Needs one item per wave (nl > 32), K <= 7 (2K codes in a dword) and (K+1)*(q-1) <= 65535; otherwise the MAC
// kernels above are used.
// ================================================================================================================

// bit j: step j adds into S1 (value 1); bit 16+j: step j adds into S2 (the other non-zero symbol).  Written with
// 0/1 flags and immediate shifts so that no per-bit constant has to live in a VGPR (a select between two literal
// masks would: 2 x 2K constants hoisted for the whole kernel).
static __device__ __forceinline__ u32 step_bits(u32 v, int j) {
  u32 is1 = v == 1u ? 1u : 0u, is2 = v > 1u ? 1u : 0u;
  asm volatile("" : "+v"(is1), "+v"(is2));              // opaque: keeps the optimiser from folding this back into selects
  return (is1 << j) | (is2 << (16 + j));
}

// Lane-conditional snapshot L1 <- S1, L2 <- S2 as an exec-masked block of in-place full-rate v_mov (hipcc would turn
// plain assignments into v_cndmask, which is far slower on gfx950: profiles/r01_microbench_exec_rate.txt).
template <int K>
static __device__ __forceinline__ void snapshot_if(bool take, u32 (&L1)[K], u32 (&L2)[K], const u32 (&S1)[K],
                                                   const u32 (&S2)[K]) {
  if (take) {
#pragma unroll
    for (int t = 0; t < K; t++) {
      asm volatile("v_mov_b32 %0, %1" : "+v"(L1[t]) : "v"(S1[t]));
      asm volatile("v_mov_b32 %0, %1" : "+v"(L2[t]) : "v"(S2[t]));
    }
  }
}

template <int K> struct TernOps;
template <>
struct TernOps<1> {
  // bit B1 of w1 set: S1[t] += W[t]; else bit B2 of w2 set: S2[t] += W[t]; else nothing.  Wave-uniform scalar tests,
  // exactly one taken branch per step, in-place full-rate adds.
  template <int B1, int B2>
  static __device__ __forceinline__ void step(u32 (&S1)[1], u32 (&S2)[1], const u32 (&W)[1], u32 w1, u32 w2) {
    asm volatile("s_bitcmp1_b32 %[wd1], %[bit1]\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitcmp1_b32 %[wd2], %[bit2]\n\t"
                 "s_cbranch_scc0 3f\n\t"
                 "v_add_u32 %[b0], %[b0], %[w0]\n\t"
                 "s_branch 3f\n"
                 "2:\n\t"
                 "v_add_u32 %[a0], %[a0], %[w0]\n\t"
                 "3:\n"
                 : [a0] "+v"(S1[0]), [b0] "+v"(S2[0])
                 : [wd1] "s"(w1), [wd2] "s"(w2), [bit1] "i"(B1), [bit2] "i"(B2), [w0] "v"(W[0])
                 : "scc");
  }
};
template <>
struct TernOps<3> {
  // bit B1 of w1 set: S1[t] += W[t]; else bit B2 of w2 set: S2[t] += W[t]; else nothing.  Wave-uniform scalar tests,
  // exactly one taken branch per step, in-place full-rate adds.
  template <int B1, int B2>
  static __device__ __forceinline__ void step(u32 (&S1)[3], u32 (&S2)[3], const u32 (&W)[3], u32 w1, u32 w2) {
    asm volatile("s_bitcmp1_b32 %[wd1], %[bit1]\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitcmp1_b32 %[wd2], %[bit2]\n\t"
                 "s_cbranch_scc0 3f\n\t"
                 "v_add_u32 %[b0], %[b0], %[w0]\n\t"
                 "v_add_u32 %[b1], %[b1], %[w1]\n\t"
                 "v_add_u32 %[b2], %[b2], %[w2]\n\t"
                 "s_branch 3f\n"
                 "2:\n\t"
                 "v_add_u32 %[a0], %[a0], %[w0]\n\t"
                 "v_add_u32 %[a1], %[a1], %[w1]\n\t"
                 "v_add_u32 %[a2], %[a2], %[w2]\n\t"
                 "3:\n"
                 : [a0] "+v"(S1[0]), [a1] "+v"(S1[1]), [a2] "+v"(S1[2]), [b0] "+v"(S2[0]), [b1] "+v"(S2[1]), [b2] "+v"(S2[2])
                 : [wd1] "s"(w1), [wd2] "s"(w2), [bit1] "i"(B1), [bit2] "i"(B2), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2])
                 : "scc");
  }
};
template <>
struct TernOps<5> {
  // bit B1 of w1 set: S1[t] += W[t]; else bit B2 of w2 set: S2[t] += W[t]; else nothing.  Wave-uniform scalar tests,
  // exactly one taken branch per step, in-place full-rate adds.
  template <int B1, int B2>
  static __device__ __forceinline__ void step(u32 (&S1)[5], u32 (&S2)[5], const u32 (&W)[5], u32 w1, u32 w2) {
    asm volatile("s_bitcmp1_b32 %[wd1], %[bit1]\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitcmp1_b32 %[wd2], %[bit2]\n\t"
                 "s_cbranch_scc0 3f\n\t"
                 "v_add_u32 %[b0], %[b0], %[w0]\n\t"
                 "v_add_u32 %[b1], %[b1], %[w1]\n\t"
                 "v_add_u32 %[b2], %[b2], %[w2]\n\t"
                 "v_add_u32 %[b3], %[b3], %[w3]\n\t"
                 "v_add_u32 %[b4], %[b4], %[w4]\n\t"
                 "s_branch 3f\n"
                 "2:\n\t"
                 "v_add_u32 %[a0], %[a0], %[w0]\n\t"
                 "v_add_u32 %[a1], %[a1], %[w1]\n\t"
                 "v_add_u32 %[a2], %[a2], %[w2]\n\t"
                 "v_add_u32 %[a3], %[a3], %[w3]\n\t"
                 "v_add_u32 %[a4], %[a4], %[w4]\n\t"
                 "3:\n"
                 : [a0] "+v"(S1[0]), [a1] "+v"(S1[1]), [a2] "+v"(S1[2]), [a3] "+v"(S1[3]), [a4] "+v"(S1[4]), [b0] "+v"(S2[0]), [b1] "+v"(S2[1]), [b2] "+v"(S2[2]), [b3] "+v"(S2[3]), [b4] "+v"(S2[4])
                 : [wd1] "s"(w1), [wd2] "s"(w2), [bit1] "i"(B1), [bit2] "i"(B2), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4])
                 : "scc");
  }
};
template <>
struct TernOps<7> {
  // bit B1 of w1 set: S1[t] += W[t]; else bit B2 of w2 set: S2[t] += W[t]; else nothing.  Wave-uniform scalar tests,
  // exactly one taken branch per step, in-place full-rate adds.
  template <int B1, int B2>
  static __device__ __forceinline__ void step(u32 (&S1)[7], u32 (&S2)[7], const u32 (&W)[7], u32 w1, u32 w2) {
    asm volatile("s_bitcmp1_b32 %[wd1], %[bit1]\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitcmp1_b32 %[wd2], %[bit2]\n\t"
                 "s_cbranch_scc0 3f\n\t"
                 "v_add_u32 %[b0], %[b0], %[w0]\n\t"
                 "v_add_u32 %[b1], %[b1], %[w1]\n\t"
                 "v_add_u32 %[b2], %[b2], %[w2]\n\t"
                 "v_add_u32 %[b3], %[b3], %[w3]\n\t"
                 "v_add_u32 %[b4], %[b4], %[w4]\n\t"
                 "v_add_u32 %[b5], %[b5], %[w5]\n\t"
                 "v_add_u32 %[b6], %[b6], %[w6]\n\t"
                 "s_branch 3f\n"
                 "2:\n\t"
                 "v_add_u32 %[a0], %[a0], %[w0]\n\t"
                 "v_add_u32 %[a1], %[a1], %[w1]\n\t"
                 "v_add_u32 %[a2], %[a2], %[w2]\n\t"
                 "v_add_u32 %[a3], %[a3], %[w3]\n\t"
                 "v_add_u32 %[a4], %[a4], %[w4]\n\t"
                 "v_add_u32 %[a5], %[a5], %[w5]\n\t"
                 "v_add_u32 %[a6], %[a6], %[w6]\n\t"
                 "3:\n"
                 : [a0] "+v"(S1[0]), [a1] "+v"(S1[1]), [a2] "+v"(S1[2]), [a3] "+v"(S1[3]), [a4] "+v"(S1[4]), [a5] "+v"(S1[5]), [a6] "+v"(S1[6]), [b0] "+v"(S2[0]), [b1] "+v"(S2[1]), [b2] "+v"(S2[2]), [b3] "+v"(S2[3]), [b4] "+v"(S2[4]), [b5] "+v"(S2[5]), [b6] "+v"(S2[6])
                 : [wd1] "s"(w1), [wd2] "s"(w2), [bit1] "i"(B1), [bit2] "i"(B2), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4]), [w5] "v"(W[5]), [w6] "v"(W[6])
                 : "scc");
  }
};
template <>
struct TernOps<9> {
  // bit B1 of w1 set: S1[t] += W[t]; else bit B2 of w2 set: S2[t] += W[t]; else nothing.  Wave-uniform scalar tests,
  // exactly one taken branch per step, in-place full-rate adds.
  template <int B1, int B2>
  static __device__ __forceinline__ void step(u32 (&S1)[9], u32 (&S2)[9], const u32 (&W)[9], u32 w1, u32 w2) {
    asm volatile("s_bitcmp1_b32 %[wd1], %[bit1]\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitcmp1_b32 %[wd2], %[bit2]\n\t"
                 "s_cbranch_scc0 3f\n\t"
                 "v_add_u32 %[b0], %[b0], %[w0]\n\t"
                 "v_add_u32 %[b1], %[b1], %[w1]\n\t"
                 "v_add_u32 %[b2], %[b2], %[w2]\n\t"
                 "v_add_u32 %[b3], %[b3], %[w3]\n\t"
                 "v_add_u32 %[b4], %[b4], %[w4]\n\t"
                 "v_add_u32 %[b5], %[b5], %[w5]\n\t"
                 "v_add_u32 %[b6], %[b6], %[w6]\n\t"
                 "v_add_u32 %[b7], %[b7], %[w7]\n\t"
                 "v_add_u32 %[b8], %[b8], %[w8]\n\t"
                 "s_branch 3f\n"
                 "2:\n\t"
                 "v_add_u32 %[a0], %[a0], %[w0]\n\t"
                 "v_add_u32 %[a1], %[a1], %[w1]\n\t"
                 "v_add_u32 %[a2], %[a2], %[w2]\n\t"
                 "v_add_u32 %[a3], %[a3], %[w3]\n\t"
                 "v_add_u32 %[a4], %[a4], %[w4]\n\t"
                 "v_add_u32 %[a5], %[a5], %[w5]\n\t"
                 "v_add_u32 %[a6], %[a6], %[w6]\n\t"
                 "v_add_u32 %[a7], %[a7], %[w7]\n\t"
                 "v_add_u32 %[a8], %[a8], %[w8]\n\t"
                 "3:\n"
                 : [a0] "+v"(S1[0]), [a1] "+v"(S1[1]), [a2] "+v"(S1[2]), [a3] "+v"(S1[3]), [a4] "+v"(S1[4]), [a5] "+v"(S1[5]), [a6] "+v"(S1[6]), [a7] "+v"(S1[7]), [a8] "+v"(S1[8]), [b0] "+v"(S2[0]), [b1] "+v"(S2[1]), [b2] "+v"(S2[2]), [b3] "+v"(S2[3]), [b4] "+v"(S2[4]), [b5] "+v"(S2[5]), [b6] "+v"(S2[6]), [b7] "+v"(S2[7]), [b8] "+v"(S2[8])
                 : [wd1] "s"(w1), [wd2] "s"(w2), [bit1] "i"(B1), [bit2] "i"(B2), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4]), [w5] "v"(W[5]), [w6] "v"(W[6]), [w7] "v"(W[7]), [w8] "v"(W[8])
                 : "scc");
  }
};
template <>
struct TernOps<11> {
  // bit B1 of w1 set: S1[t] += W[t]; else bit B2 of w2 set: S2[t] += W[t]; else nothing.  Wave-uniform scalar tests,
  // exactly one taken branch per step, in-place full-rate adds.
  template <int B1, int B2>
  static __device__ __forceinline__ void step(u32 (&S1)[11], u32 (&S2)[11], const u32 (&W)[11], u32 w1, u32 w2) {
    asm volatile("s_bitcmp1_b32 %[wd1], %[bit1]\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitcmp1_b32 %[wd2], %[bit2]\n\t"
                 "s_cbranch_scc0 3f\n\t"
                 "v_add_u32 %[b0], %[b0], %[w0]\n\t"
                 "v_add_u32 %[b1], %[b1], %[w1]\n\t"
                 "v_add_u32 %[b2], %[b2], %[w2]\n\t"
                 "v_add_u32 %[b3], %[b3], %[w3]\n\t"
                 "v_add_u32 %[b4], %[b4], %[w4]\n\t"
                 "v_add_u32 %[b5], %[b5], %[w5]\n\t"
                 "v_add_u32 %[b6], %[b6], %[w6]\n\t"
                 "v_add_u32 %[b7], %[b7], %[w7]\n\t"
                 "v_add_u32 %[b8], %[b8], %[w8]\n\t"
                 "v_add_u32 %[b9], %[b9], %[w9]\n\t"
                 "v_add_u32 %[b10], %[b10], %[w10]\n\t"
                 "s_branch 3f\n"
                 "2:\n\t"
                 "v_add_u32 %[a0], %[a0], %[w0]\n\t"
                 "v_add_u32 %[a1], %[a1], %[w1]\n\t"
                 "v_add_u32 %[a2], %[a2], %[w2]\n\t"
                 "v_add_u32 %[a3], %[a3], %[w3]\n\t"
                 "v_add_u32 %[a4], %[a4], %[w4]\n\t"
                 "v_add_u32 %[a5], %[a5], %[w5]\n\t"
                 "v_add_u32 %[a6], %[a6], %[w6]\n\t"
                 "v_add_u32 %[a7], %[a7], %[w7]\n\t"
                 "v_add_u32 %[a8], %[a8], %[w8]\n\t"
                 "v_add_u32 %[a9], %[a9], %[w9]\n\t"
                 "v_add_u32 %[a10], %[a10], %[w10]\n\t"
                 "3:\n"
                 : [a0] "+v"(S1[0]), [a1] "+v"(S1[1]), [a2] "+v"(S1[2]), [a3] "+v"(S1[3]), [a4] "+v"(S1[4]), [a5] "+v"(S1[5]), [a6] "+v"(S1[6]), [a7] "+v"(S1[7]), [a8] "+v"(S1[8]), [a9] "+v"(S1[9]), [a10] "+v"(S1[10]), [b0] "+v"(S2[0]), [b1] "+v"(S2[1]), [b2] "+v"(S2[2]), [b3] "+v"(S2[3]), [b4] "+v"(S2[4]), [b5] "+v"(S2[5]), [b6] "+v"(S2[6]), [b7] "+v"(S2[7]), [b8] "+v"(S2[8]), [b9] "+v"(S2[9]), [b10] "+v"(S2[10])
                 : [wd1] "s"(w1), [wd2] "s"(w2), [bit1] "i"(B1), [bit2] "i"(B2), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4]), [w5] "v"(W[5]), [w6] "v"(W[6]), [w7] "v"(W[7]), [w8] "v"(W[8]), [w9] "v"(W[9]), [w10] "v"(W[10])
                 : "scc");
  }
};
template <>
struct TernOps<13> {
  // bit B1 of w1 set: S1[t] += W[t]; else bit B2 of w2 set: S2[t] += W[t]; else nothing.  Wave-uniform scalar tests,
  // exactly one taken branch per step, in-place full-rate adds.
  template <int B1, int B2>
  static __device__ __forceinline__ void step(u32 (&S1)[13], u32 (&S2)[13], const u32 (&W)[13], u32 w1, u32 w2) {
    asm volatile("s_bitcmp1_b32 %[wd1], %[bit1]\n\t"
                 "s_cbranch_scc1 2f\n\t"
                 "s_bitcmp1_b32 %[wd2], %[bit2]\n\t"
                 "s_cbranch_scc0 3f\n\t"
                 "v_add_u32 %[b0], %[b0], %[w0]\n\t"
                 "v_add_u32 %[b1], %[b1], %[w1]\n\t"
                 "v_add_u32 %[b2], %[b2], %[w2]\n\t"
                 "v_add_u32 %[b3], %[b3], %[w3]\n\t"
                 "v_add_u32 %[b4], %[b4], %[w4]\n\t"
                 "v_add_u32 %[b5], %[b5], %[w5]\n\t"
                 "v_add_u32 %[b6], %[b6], %[w6]\n\t"
                 "v_add_u32 %[b7], %[b7], %[w7]\n\t"
                 "v_add_u32 %[b8], %[b8], %[w8]\n\t"
                 "v_add_u32 %[b9], %[b9], %[w9]\n\t"
                 "v_add_u32 %[b10], %[b10], %[w10]\n\t"
                 "v_add_u32 %[b11], %[b11], %[w11]\n\t"
                 "v_add_u32 %[b12], %[b12], %[w12]\n\t"
                 "s_branch 3f\n"
                 "2:\n\t"
                 "v_add_u32 %[a0], %[a0], %[w0]\n\t"
                 "v_add_u32 %[a1], %[a1], %[w1]\n\t"
                 "v_add_u32 %[a2], %[a2], %[w2]\n\t"
                 "v_add_u32 %[a3], %[a3], %[w3]\n\t"
                 "v_add_u32 %[a4], %[a4], %[w4]\n\t"
                 "v_add_u32 %[a5], %[a5], %[w5]\n\t"
                 "v_add_u32 %[a6], %[a6], %[w6]\n\t"
                 "v_add_u32 %[a7], %[a7], %[w7]\n\t"
                 "v_add_u32 %[a8], %[a8], %[w8]\n\t"
                 "v_add_u32 %[a9], %[a9], %[w9]\n\t"
                 "v_add_u32 %[a10], %[a10], %[w10]\n\t"
                 "v_add_u32 %[a11], %[a11], %[w11]\n\t"
                 "v_add_u32 %[a12], %[a12], %[w12]\n\t"
                 "3:\n"
                 : [a0] "+v"(S1[0]), [a1] "+v"(S1[1]), [a2] "+v"(S1[2]), [a3] "+v"(S1[3]), [a4] "+v"(S1[4]), [a5] "+v"(S1[5]), [a6] "+v"(S1[6]), [a7] "+v"(S1[7]), [a8] "+v"(S1[8]), [a9] "+v"(S1[9]), [a10] "+v"(S1[10]), [a11] "+v"(S1[11]), [a12] "+v"(S1[12]), [b0] "+v"(S2[0]), [b1] "+v"(S2[1]), [b2] "+v"(S2[2]), [b3] "+v"(S2[3]), [b4] "+v"(S2[4]), [b5] "+v"(S2[5]), [b6] "+v"(S2[6]), [b7] "+v"(S2[7]), [b8] "+v"(S2[8]), [b9] "+v"(S2[9]), [b10] "+v"(S2[10]), [b11] "+v"(S2[11]), [b12] "+v"(S2[12])
                 : [wd1] "s"(w1), [wd2] "s"(w2), [bit1] "i"(B1), [bit2] "i"(B2), [w0] "v"(W[0]), [w1] "v"(W[1]), [w2] "v"(W[2]), [w3] "v"(W[3]), [w4] "v"(W[4]), [w5] "v"(W[5]), [w6] "v"(W[6]), [w7] "v"(W[7]), [w8] "v"(W[8]), [w9] "v"(W[9]), [w10] "v"(W[10]), [w11] "v"(W[11]), [w12] "v"(W[12])
                 : "scc");
  }
};

// Steps J .. 2K-1 of one block (compile-time recursion so every bit index / register index is an immediate).
// word: bit j = "step j adds into S1", bit 16+j = "step j adds into S2".  ME: mask both sets every ME steps
// (0 = never: exact small sums).  At most ME additions of values < q land on a masked field between masks.
template <int K, int ME, int J>
static __device__ __forceinline__ void tern_steps(u32 (&S1)[K], u32 (&S2)[K], u32 (&WE)[K], u32 (&WO)[K],
                                                  const uint2 (&nw)[K], u32 word, u32 fmask) {
  if constexpr (J < 2 * K) {
    if constexpr (ME > 0 && J % (ME > 0 ? ME : 1) == 0) {
#pragma unroll
      for (int t = 0; t < K; t++) { S1[t] &= fmask; S2[t] &= fmask; }
    }
    constexpr int s = J >> 1;
    u32 W[K];
#pragma unroll
    for (int t = 0; t < K; t++) W[t] = (J & 1) ? WO[(t - s + K) % K] : WE[(t - s + K) % K];
    TernOps<K>::template step<J, 16 + J>(S1, S2, W, word, word);
    if constexpr ((J & 1) != 0) { WE[K - 1 - s] = nw[s].x; WO[K - 1 - s] = nw[s].y; }
    tern_steps<K, ME, J + 1>(S1, S2, WE, WO, nw, word, fmask);
  }
}

template <int K, int ME, bool WL>
static __device__ __forceinline__ void tern_core(const uint2 *__restrict__ eo, const u32 *__restrict__ codes,
                                                 const Geom &g, int sub, u32 fmask,
                                                 u32 (&S1)[K], u32 (&S2)[K], u32 (&L1)[K], u32 (&L2)[K]) {
  u32 WE[K], WO[K];
  const uint2 *nb = eo + (K * sub + g.off);
#pragma unroll
  for (int x = 0; x < K; x++) { uint2 v = nb[x]; WE[x] = v.x; WO[x] = v.y; }
#pragma unroll
  for (int t = 0; t < K; t++) { S1[t] = 0; S2[t] = 0; L1[t] = 0; L2[t] = 0; }
  const int nblk = g.nl;
  u32 word = __builtin_amdgcn_readfirstlane(codes[0]);
  for (int m = 0; m < nblk; m++) {
    const u32 next_raw = codes[m + 1 < nblk ? m + 1 : m];
    if constexpr (WL) snapshot_if<K>(m == sub, L1, L2, S1, S2);
    uint2 nw[K];
#pragma unroll
    for (int s = 0; s < K; s++) nw[s] = nb[-1 - s];
    tern_steps<K, ME, 0>(S1, S2, WE, WO, nw, word, fmask);
    word = __builtin_amdgcn_readfirstlane(next_raw);
    nb -= K;
  }
}

// T = S1 + c*S2 per 16-bit field, as a u16 pair (mod 2^16 from here on).  NEG: c = -1 (mod q), else c = 2.
template <bool NEG>
static __device__ __forceinline__ u16x2 tern_combine(u32 s1, u32 s2, u32 fmask, u32 qq) {
  s1 &= fmask; s2 &= fmask;
  return as_pair(NEG ? s1 + (qq - s2) : s1 + (s2 << 1));
}

// Finish one ternary-stepped product: remainder / quotient pairs like product_split.
//   av: the lane's own 2K stepping-operand values (numeric, as u16 pairs) for the in-block triangle.
// Which lanes may store a whole block of 2K outputs without bounds checks, and which one holds the row's tail.
struct StorePlan {
  bool full, tail; int nv;      // nv: number of valid outputs in the tail lane (wave-uniform)
};
template <int K>
static __device__ __forceinline__ StorePlan store_plan(const Geom &g, int sub, bool valid) {
  StorePlan sp;
  sp.nv = g.N - 2 * K * (g.nl - 1);
  sp.full = valid && (sub < g.nl - 1 || sp.nv == 2 * K);
  sp.tail = valid && !sp.full;
  return sp;
}
// Store output pair t of this lane into its row (`lane_row` already points at the lane's first output).
template <class OutT>
static __device__ __forceinline__ void store_pair(OutT *lane_row, const StorePlan &sp, int t, u16x2 v) {
  if (sp.full) {
    lane_row[2 * t] = (OutT)v.x; lane_row[2 * t + 1] = (OutT)v.y;
  } else if (sp.tail) {
    if (2 * t < sp.nv) lane_row[2 * t] = (OutT)v.x;        // wave-uniform tests
    if (2 * t + 1 < sp.nv) lane_row[2 * t + 1] = (OutT)v.y;
  }
}

// Runs one per-item-stepped product and hands each finished pair to `emit(t, rem_pair, quot_pair)` right away.
// av(s): the lane's s-th pair of the stepping operand (numeric), only evaluated for the triangle after the main loop.
template <int K, int ME, bool NEG, class AV, class Emit>
static __device__ __forceinline__ void tern_product_split(const uint2 *eo, const u32 *codes, AV av,
                                                          const Geom &g, int sub, bool want_quot, u32 mod, Emit emit) {
  constexpr bool POW2 = ME > 0;
  const u32 fmask = POW2 ? (mod - 1) * 0x00010001u : 0xFFFFFFFFu;
  const u32 qq = mod * 0x00010001u;
  u32 S1[K], S2[K], L1[K], L2[K];
  if (want_quot) tern_core<K, ME, true>(eo, codes, g, sub, fmask, S1, S2, L1, L2);
  else tern_core<K, ME, false>(eo, codes, g, sub, fmask, S1, S2, L1, L2);
  u16x2 low[K];
#pragma unroll
  for (int t = 0; t < K; t++) low[t] = tern_combine<NEG>(L1[t], L2[t], fmask, qq);
  if (want_quot) {                                   // in-block triangle
    u32 ZE[K], ZO[K];
#pragma unroll
    for (int x = 0; x < K; x++) { uint2 v = eo[g.off + x]; ZE[x] = v.x; ZO[x] = v.y; }
    ZO[0] &= 0xFFFF0000u;
#pragma unroll
    for (int s = 0; s < K; s++) {
      const u16x2 ap = as_pair(av(s));
#pragma unroll
      for (int t = s; t < K; t++) {
        low[t] = ap.xx * as_pair(ZE[t - s]) + low[t];
        low[t] = ap.yy * as_pair(ZO[t - s]) + low[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < K; t++) {
    const u16x2 T = tern_combine<NEG>(S1[t], S2[t], fmask, qq);
    u16x2 qv = (u16x2){0, 0};
    if (want_quot) {
      const u16x2 hi = T - low[t];
      if (POW2) {
        qv = ((u16x2){0, 0} - hi) & (u16)(mod - 1);
      } else {
        const u32 h0 = mod_small(hi.x, mod), h1 = mod_small(hi.y, mod);
        qv = (u16x2){(u16)(h0 ? mod - h0 : 0), (u16)(h1 ? mod - h1 : 0)};
      }
    }
    const u16x2 rv = POW2 ? T : (u16x2){(u16)mod_small(T.x, mod), (u16)mod_small(T.y, mod)};   // POW2: caller masks
    emit(t, rv, qv);
  }
}

// Load the lane's block of the stepping operand: numeric u16 pairs for the triangle + the block's code word.
template <int K, class F>
static __device__ __forceinline__ u32 load_block(F val, int N, int sub, u32 (&av)[K]) {
  u32 word = 0;
#pragma unroll
  for (int t = 0; t < K; t++) {
    const int k = 2 * K * sub + 2 * t;
    const int k0 = k < N ? k : N - 1, k1 = k + 1 < N ? k + 1 : N - 1;          // clamped, then zeroed
    const u32 v0 = k < N ? (val(k0) & 0xFFFFu) : 0u, v1 = k + 1 < N ? (val(k1) & 0xFFFFu) : 0u;
    av[t] = v0 | (v1 << 16);
    word |= step_bits(v0, 2 * t) | step_bits(v1, 2 * t + 1);
  }
  return word;
}

// encryptBits on the add path: stepping operand r in {0,1,2}, window h (shared).
template <int K, int ME>
__global__ __launch_bounds__(BLOCK_THREADS, 7) void k_encrypt_t(Geom g, u32 q, const u16 *__restrict__ h,
                                                             const uint8_t *__restrict__ r,
                                                             const uint8_t *__restrict__ m, long B,
                                                             u16 *__restrict__ e, u16 *__restrict__ quotE) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint2 *eo_h = (uint2 *)lds;
  const LaneId L = lane_id(g);
  u32 *codes = (u32 *)(lds + (size_t)g.eo_len * 8) + (size_t)L.wave * g.nl;
  build_eo(eo_h, g, ValU16{h}, threadIdx.x, BLOCK_THREADS);
  __syncthreads();
  const bool want_quot = quotE != nullptr;
  for (long item = (long)blockIdx.x * WAVES_PER_BLOCK + L.wave; item < B; item += (long)gridDim.x * WAVES_PER_BLOCK) {
    int sub = L.sub, N = g.N;                            // re-materialised per iteration: keeps the glue out of LICM's reach
    asm volatile("" : "+v"(sub), "+s"(N));
    const long row = item * N;
    {
      u32 av[K];
      const u32 word = load_block<K>(ValU8{r + row}, N, sub, av);
      if (L.active) codes[sub] = word;
    }
    wave_lds_fence();
    // everything below runs behind the hot loop; `sub2` is a fresh opaque copy so that none of its index arithmetic
    // is CSE'd with the pre-loop block load and kept live (spilled) across the loop
    auto r_pair = [&](int t) -> u32 {                                     // reloaded behind the hot loop (L1/L2 hit)
      int sub2 = sub;
      asm volatile("" : "+v"(sub2));
      const uint8_t *rr = r + row;
      const int k = 2 * K * sub2 + 2 * t;
      const int k0 = k < N ? k : N - 1, k1 = k + 1 < N ? k + 1 : N - 1;
      const u32 v0 = k < N ? rr[k0] : 0u, v1 = k + 1 < N ? rr[k1] : 0u;
      return v0 | (v1 << 16);
    };
    tern_product_split<K, ME, false>(eo_h, codes, r_pair, g, sub, want_quot, q, [&](int t, u16x2 rv, u16x2 qv) {
      int sub2 = sub;
      asm volatile("" : "+v"(sub2));
      const StorePlan sp = store_plan<K>(g, sub2, L.active);
      const long lane0 = row + 2 * K * sub2;
      const int k = 2 * K * sub2 + 2 * t;
      const int k0 = k < N ? k : N - 1, k1 = k + 1 < N ? k + 1 : N - 1;
      const uint8_t *mr = m + row;
      const u16x2 add = {(u16)mr[k0], (u16)mr[k1]};                         // out-of-row lanes never store
      store_pair(e + lane0, sp, t, (rv + add) & (u16)(q - 1));
      if (want_quot) store_pair(quotE + lane0, sp, t, qv);
    });
    wave_lds_fence();
  }
}

// decryptBits on the add path: product 1 steps over f (shared, codes built once) with a per-item window of e;
// product 2 steps over the lifted message (per item, in registers) with the shared window of fp.  Needs p == 3.
template <int K, int ME>
__global__ __launch_bounds__(BLOCK_THREADS) void k_decrypt_t(Geom g, u32 q, u32 p,
                                                             const int8_t *__restrict__ f,
                                                             const uint8_t *__restrict__ fp,
                                                             const u16 *__restrict__ e, long B,
                                                             uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                             u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint2 *eo_fp = (uint2 *)lds;
  u32 *codes_f = (u32 *)(lds + (size_t)g.eo_len * 8);
  const LaneId L = lane_id(g);
  const size_t per_wave = (size_t)g.eo_len * 8 + (size_t)g.nl * 4;
  unsigned char *wbase = lds + (size_t)g.eo_len * 8 + (size_t)g.nl * 4 + (size_t)L.wave * per_wave;
  uint2 *eo_e = (uint2 *)wbase;
  u32 *codes_b = (u32 *)(wbase + (size_t)g.eo_len * 8);
  build_eo(eo_fp, g, ValU8{fp}, threadIdx.x, BLOCK_THREADS);
  u32 av_f[K];                                           // this lane's block of f, the same for every item
  {
    const u32 wf = load_block<K>(ValTernary{f, q - 1}, g.N, L.sub, av_f);
    if (L.wave == 0 && L.active) codes_f[L.sub] = wf;
  }
  __syncthreads();
  const bool want_q1 = quot1 != nullptr, want_q2 = quot2 != nullptr;
  for (long item = (long)blockIdx.x * WAVES_PER_BLOCK + L.wave; item < B; item += (long)gridDim.x * WAVES_PER_BLOCK) {
    const long row = item * g.N;
    if (L.active) build_eo(eo_e, g, ValU16{e + row}, L.sub, g.nl);
    wave_lds_fence();
    const StorePlan sp = store_plan<K>(g, L.sub, L.active);
    const long lane0 = row + 2 * K * L.sub;
    u32 av_b[K], wb = 0;
    // remainder1 / quotient1 stored pair by pair; centred lift, index.js:117 verbatim -> second stepping operand
    tern_product_split<K, ME, true>(eo_e, codes_f, [&](int t) { return av_f[t]; }, g, L.sub, want_q1, q, [&](int t, u16x2 rv, u16x2 qv) {
      rv = rv & (u16)(q - 1);
      if (rem1) store_pair(rem1 + lane0, sp, t, rv);
      if (want_q1) store_pair(quot1 + lane0, sp, t, qv);
      const int k = 2 * K * L.sub + 2 * t;
      const u32 x0 = rv.x, x1 = rv.y;
      u32 b0 = mod_small(2 * x0 > q ? x0 + 1 : x0, p), b1 = mod_small(2 * x1 > q ? x1 + 1 : x1, p);
      b0 = k < g.N ? b0 : 0; b1 = k + 1 < g.N ? b1 : 0;
      av_b[t] = b0 | (b1 << 16);
      wb |= step_bits(b0, 2 * t) | step_bits(b1, 2 * t + 1);
    });
    if (L.active) codes_b[L.sub] = wb;
    wave_lds_fence();
    tern_product_split<K, 0, false>(eo_fp, codes_b, [&](int t) { return av_b[t]; }, g, L.sub, want_q2, p, [&](int t, u16x2 rv, u16x2 qv) {
      store_pair(value + lane0, sp, t, rv);
      if (want_q2) store_pair(quot2 + lane0, sp, t, qv);
    });
    wave_lds_fence();
  }
}


// ================================================================================================================
// Shared-stepping add path (decrypt).  Both products of decryptBits can step over a SHARED key operand (f, then fp),
// so every wave of the launch follows the same step masks.  That allows two items per wave (32 lanes x K pairs each,
// K = 9 / 11 / 13) under one scalar control stream -- the scalar unit, not the VALU, is what limits the add path
// (profiles/r01_microbench_step_rate.txt).  The per-item operand (e, then the lifted message) is the window; it is kept
// in LDS as ONE cyclic array of aligned pairs E[u] = (bc[2u], bc[2u+1]); the odd-aligned pairs are derived on the fly,
// O[u] = alignbit(E[u], E[u-1], 16).  N must be odd (so that the cyclic wrap turns aligned pairs into odd-aligned ones).
// ================================================================================================================

static __device__ __forceinline__ u32 odd_pair(u32 e_u, u32 e_um1) { return __builtin_amdgcn_alignbit(e_u, e_um1, 16); }

// Fill this item's cyclic pair array from the K aligned pairs P[t] = (x[2v], x[(2v+1) mod N]), v = K*sub + t, held in
// registers by the item's lanes.  E points at logical entry u = -off (one spare entry sits in front of it, one dummy
// slot behind the array absorbs writes of lanes that have nothing to contribute -- no per-element predication).
//   phase 1: E[u] = P[u] for u in [0, H), H = (N+1)/2 (the aligned pairs, the last one wrapping to x[0])
//   phase 2: every other entry is either another aligned pair (2u mod N even) or the odd-aligned pair
//            (x[j], x[j+1]) = (hi(P[(j-1)/2]), lo(P[(j+1)/2])), j = 2u mod N, read back from phase 1's region
template <int K>
static __device__ __forceinline__ void build_cyclic_pairs(u32 *E, const Geom &g, int sub, bool active, const u32 (&P)[K],
                                                          bool patch_wrap, u32 x0) {
  const int N = g.N, H = (N + 1) >> 1, off = g.off, top = K * g.nl;
  const int dummy = top + off;                          // one slot past the last real entry
#pragma unroll
  for (int t = 0; t < K; t++) {
    const int v = K * sub + t;
    E[(active && v < H) ? v + off : dummy] = P[t];
  }
  wave_lds_fence();
  if (patch_wrap && active && sub == 0) ((u16 *)E)[2 * (H - 1 + off) + 1] = (u16)x0;   // P[H-1] = (x[N-1], x[0])
  wave_lds_fence();
  const int below = off + 1, total = below + (top - H);  // entries u in [-off-1, -1] and [H, top)
  for (int idx = sub; idx < total; idx += g.nl) {
    const int u = idx < below ? idx - below : H + (idx - below);
    int j = 2 * u;                                       // reduce 2u into [0, N)
    j += j < 0 ? N : 0; j += j < 0 ? N : 0; j -= j >= N ? N : 0;
    const int hi_src = (j + 1) >> 1, lo_src = j >> 1;    // j even: both = j/2 (aligned pair); j odd: neighbours
    const u32 a = E[hi_src + off], b = E[lo_src + off];
    const u32 val = (j & 1) ? odd_pair(a, b) : a;
    E[active ? u + off : dummy] = val;
  }
  wave_lds_fence();
}

template <int K, int ME, int J>
static __device__ __forceinline__ void shared_steps(u32 (&S1)[K], u32 (&S2)[K], u32 (&WE)[K], u32 (&WO)[K],
                                                    const u32 (&nw)[K + 1], u32 ones, u32 twos, u32 fmask) {
  if constexpr (J < 2 * K) {
    if constexpr (ME > 0 && J % (ME > 0 ? ME : 1) == 0) {
#pragma unroll
      for (int t = 0; t < K; t++) { S1[t] &= fmask; S2[t] &= fmask; }
    }
    constexpr int s = J >> 1;
    u32 W[K];
#pragma unroll
    for (int t = 0; t < K; t++) W[t] = (J & 1) ? WO[(t - s + K) % K] : WE[(t - s + K) % K];
    TernOps<K>::template step<J, J>(S1, S2, W, ones, twos);
    if constexpr ((J & 1) != 0) { WE[K - 1 - s] = nw[s]; WO[K - 1 - s] = odd_pair(nw[s], nw[s + 1]); }
    shared_steps<K, ME, J + 1>(S1, S2, WE, WO, nw, ones, twos, fmask);
  }
}

// masks: one uint2 per block (x: steps adding into S1, y: steps adding into S2), identical for every item.
template <int K, int ME, bool WL>
static __device__ __forceinline__ void shared_core(const u32 *__restrict__ E, const uint2 *__restrict__ masks,
                                                   const Geom &g, int sub, u32 fmask,
                                                   u32 (&S1)[K], u32 (&S2)[K], u32 (&L1)[K], u32 (&L2)[K]) {
  u32 WE[K], WO[K];
  const u32 *nb = E + (K * sub + g.off);
  {
    u32 prev = nb[-1];
#pragma unroll
    for (int x = 0; x < K; x++) { const u32 v = nb[x]; WE[x] = v; WO[x] = odd_pair(v, prev); prev = v; }
  }
#pragma unroll
  for (int t = 0; t < K; t++) { S1[t] = 0; S2[t] = 0; L1[t] = 0; L2[t] = 0; }
  const int nblk = g.nl;
  uint2 mk = masks[0];
  for (int m = 0; m < nblk; m++) {
    const u32 ones = __builtin_amdgcn_readfirstlane(mk.x), twos = __builtin_amdgcn_readfirstlane(mk.y);
    mk = masks[m + 1 < nblk ? m + 1 : m];
    if constexpr (WL) snapshot_if<K>(m == sub, L1, L2, S1, S2);
    u32 nw[K + 1];
#pragma unroll
    for (int s = 0; s <= K; s++) nw[s] = nb[-1 - s];
    shared_steps<K, ME, 0>(S1, S2, WE, WO, nw, ones, twos, fmask);
    nb -= K;
  }
}

// Finish a shared-stepped product (same contract as tern_product_split; av = the lane's block of the stepping operand).
// Runs one shared-stepped product and hands each finished pair to `emit(t, rem_pair, quot_pair)` right away (so no
// result arrays stay live: register pressure is what limits this kernel's occupancy).
template <int K, int ME, bool NEG, class Emit>
static __device__ __forceinline__ void shared_product_split(const u32 *E, const uint2 *masks, const u32 *av,
                                                            const Geom &g, int sub, bool want_quot, u32 mod, Emit emit) {
  constexpr bool POW2 = ME > 0;
  const u32 fmask = POW2 ? (mod - 1) * 0x00010001u : 0xFFFFFFFFu;
  const u32 qq = mod * 0x00010001u;
  u32 S1[K], S2[K], L1[K], L2[K];
  if (want_quot) shared_core<K, ME, true>(E, masks, g, sub, fmask, S1, S2, L1, L2);
  else shared_core<K, ME, false>(E, masks, g, sub, fmask, S1, S2, L1, L2);
  // fold the snapshots into "low" right away (frees L2), then add the in-block triangle
  u16x2 low[K];
#pragma unroll
  for (int t = 0; t < K; t++) low[t] = tern_combine<NEG>(L1[t], L2[t], fmask, qq);
  if (want_quot) {
    u32 ZE[K], ZO[K];
    u32 prev = 0;                                    // the linear product has no coefficient before index 0
#pragma unroll
    for (int x = 0; x < K; x++) { const u32 v = E[g.off + x]; ZE[x] = v; ZO[x] = odd_pair(v, prev); prev = v; }
#pragma unroll
    for (int s = 0; s < K; s++) {
      const u16x2 ap = as_pair(av[s]);
#pragma unroll
      for (int t = s; t < K; t++) {
        low[t] = ap.xx * as_pair(ZE[t - s]) + low[t];
        low[t] = ap.yy * as_pair(ZO[t - s]) + low[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < K; t++) {
    const u16x2 T = tern_combine<NEG>(S1[t], S2[t], fmask, qq);
    u16x2 qv = (u16x2){0, 0};
    if (want_quot) {
      const u16x2 hi = T - low[t];
      if (POW2) {
        qv = ((u16x2){0, 0} - hi) & (u16)(mod - 1);
      } else {
        const u32 h0 = mod_small(hi.x, mod), h1 = mod_small(hi.y, mod);
        qv = (u16x2){(u16)(h0 ? mod - h0 : 0), (u16)(h1 ? mod - h1 : 0)};
      }
    }
    const u16x2 rv = POW2 ? (T & (u16)(mod - 1)) : (u16x2){(u16)mod_small(T.x, mod), (u16)mod_small(T.y, mod)};
    emit(t, rv, qv);
  }
}


// ---- ternary x ternary product on v_dot8_u32_u4 (decrypt's c = fp * b mod 3) ---------------------------------------
// Both operands are in {0,1,2}: 8 multiply-accumulates per instruction, no branches, exact sums (<= 4N).  Layout for
// this phase: lane l owns the 32 outputs k = 32 l + t.  A8r[I] (shared, built once per workgroup) packs fp[8I+7-i] in
// nibble i; the per-item operand b is a nibble stream in LDS (coefficient j at nibble j + 8*nblk, extended cyclically
// below 0), and FB(t, I) = the forward window b[k-8I-7 .. k-8I] is cut out of two stream dwords with v_alignbit; going
// to the next block shifts all windows by 8 slots, so only 8 of the 32 are recomputed.  The low half for the quotient
// is the accumulator snapshot before block 4 l plus a 4-block correction against the zero-extended stream.
// tools/dot8_model.py is the executable specification (checked against a direct convolution).
static __device__ __forceinline__ u32 funnel(u32 hi, u32 lo, int sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// nibble stream of b from the K pairs per lane of the 32-lane product-1 layout (P[t] = (b[2v], b[2v+1]), v = K sub + t)
template <int K>
static __device__ __forceinline__ void build_nibble_stream(u32 *dwp, int N, int nblk, int sub, bool active,
                                                           const u32 (&P)[K]) {
  if (active) {
    unsigned char *by = (unsigned char *)dwp + 4 * nblk + K * sub;       // coefficient 0 sits at dword nblk
#pragma unroll
    for (int t = 0; t < K; t++) by[t] = (unsigned char)((P[t] & 0xFu) | (((P[t] >> 16) & 0xFu) << 4));
  }
  wave_lds_fence();
  const int a = N >> 3, r4 = 4 * (N & 7);                                 // N nibbles = a dwords + r nibbles
  for (int w = sub; w < nblk; w += 32) {                                 // coefficient j < 0 is b[j + N]
    const u32 lo = dwp[w + a], hi = dwp[w + a + 1];
    if (active) dwp[w] = r4 ? funnel(hi, lo, r4) : lo;
  }
  wave_lds_fence();
}

template <bool WL>
static __device__ __forceinline__ void dot8_core(const u32 *__restrict__ dwp, const u32 *__restrict__ a8, int nblk, int l,
                                                 u32 (&acc)[32], u32 (&snap)[32]) {
  u32 S[32];
  const u32 *base = dwp + 4 * l + nblk - 1;
  u32 Y;
  {
    u32 D[5];
#pragma unroll
    for (int g = 0; g < 5; g++) D[g] = base[g];
#pragma unroll
    for (int t = 0; t < 32; t++) {
      const int g = (t + 1) >> 3, ph = (t + 1) & 7;
      S[t] = ph ? funnel(D[g + 1 < 5 ? g + 1 : 4], D[g], 4 * ph) : D[g];
    }
    Y = D[0];
  }
#pragma unroll
  for (int t = 0; t < 32; t++) { acc[t] = 0; snap[t] = 0; }
  const u32 *xp = base - 1;                                               // block I reads xp[-I]
  for (int u = 0; u < (nblk >> 2); u++) {
    if constexpr (WL) {
      if (u == l) {
#pragma unroll
        for (int t = 0; t < 32; t++) asm volatile("v_mov_b32 %0, %1" : "+v"(snap[t]) : "v"(acc[t]));
      }
    }
    const uint4 av = *(const uint4 *)(a8 + 4 * u);
    const u32 A[4] = {av.x, av.y, av.z, av.w};
    u32 X[4];
#pragma unroll
    for (int v = 0; v < 4; v++) X[v] = xp[-(4 * u + v)];
#pragma unroll
    for (int v = 0; v < 4; v++) {
#pragma unroll
      for (int t = 0; t < 32; t++) acc[t] = __builtin_amdgcn_udot8(A[v], S[(t - 8 * v) & 31], acc[t], false);
#pragma unroll
      for (int tp = 0; tp < 7; tp++) S[(tp - 8 * (v + 1)) & 31] = funnel(Y, X[v], 4 * (tp + 1));
      S[(7 - 8 * (v + 1)) & 31] = Y;
      Y = X[v];
    }
  }
}

// c = fp * b mod 3 for one item per 32-lane half: value / quotient2 rows stored directly.
template <bool WQ>
static __device__ __forceinline__ void dot8_product_mod3(const u32 *dwp, const u32 *a8, int N, int nblk, int l, bool valid,
                                                         int lanes, uint8_t *__restrict__ value_row,
                                                         uint8_t *__restrict__ quot_row) {
  u32 acc[32], low[32];
  dot8_core<WQ>(dwp, a8, nblk, l, acc, low);
  if (WQ) {                                             // low += in-block part, against the zero-extended stream
    u32 ZD[9];
#pragma unroll
    for (int g = 0; g < 4; g++) ZD[g] = 0;
#pragma unroll
    for (int g = 0; g < 5; g++) ZD[4 + g] = dwp[nblk + g];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const u32 a = a8[4 * l + d];
#pragma unroll
      for (int t = 0; t < 32; t++) {
        const int g = ((t + 1) >> 3) + 3 - d, ph = (t + 1) & 7;
        const u32 w = ph ? funnel(ZD[g + 1], ZD[g], 4 * ph) : ZD[g];
        low[t] = __builtin_amdgcn_udot8(a, w, low[t], false);
      }
    }
  }
  const int nv = N - 32 * (lanes - 1);                  // valid outputs of the last lane (wave-uniform)
  const bool full = valid && l < lanes - 1, tail = valid && l == lanes - 1;
  uint8_t *vr = value_row + 32 * l, *qr = quot_row + 32 * l;
#pragma unroll
  for (int t = 0; t < 32; t++) {
    const u32 T = acc[t];
    const u32 rv = T % 3u;
    u32 qv = 0;
    if (WQ) { const u32 h = (T - low[t]) % 3u; qv = h ? 3u - h : 0u; }
    if (full || (tail && t < nv)) {
      vr[t] = (uint8_t)rv;
      if (WQ) qr[t] = (uint8_t)qv;
    }
  }
}

// The lane's block of a shared stepping operand: numeric pairs + the block's two step masks.
template <int K, class F>
static __device__ __forceinline__ uint2 load_block_masks(F val, const Geom &g, int sub, u32 (&av)[K]) {
  uint2 mk = make_uint2(0u, 0u);
#pragma unroll
  for (int t = 0; t < K; t++) {
    const int k = 2 * K * sub + 2 * t;
    const u32 v0 = k < g.N ? (val(k) & 0xFFFFu) : 0u, v1 = k + 1 < g.N ? (val(k + 1) & 0xFFFFu) : 0u;
    av[t] = v0 | (v1 << 16);
    u32 a1 = v0 == 1u ? 1u : 0u, a2 = v0 > 1u ? 1u : 0u, b1 = v1 == 1u ? 1u : 0u, b2 = v1 > 1u ? 1u : 0u;
    asm volatile("" : "+v"(a1), "+v"(a2), "+v"(b1), "+v"(b2));   // see step_bits
    mk.x |= (a1 << (2 * t)) | (b1 << (2 * t + 1));
    mk.y |= (a2 << (2 * t)) | (b2 << (2 * t + 1));
  }
  return mk;
}

// decryptBits with both products stepping over the shared key (f, then fp); two items per wave.  p must be 3.
template <int K, int ME, bool D8>
__global__ __launch_bounds__(BLOCK_THREADS, 4) void k_decrypt_s(Geom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                             const uint8_t *__restrict__ fp,
                                                             const u16 *__restrict__ e, long B,
                                                             uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                             u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  uint2 *masks_f = (uint2 *)lds;
  uint2 *masks_fp = masks_f + g.nl;
  u32 *blk_f = (u32 *)(masks_fp + g.nl);                                // [nl][K] numeric pairs of f (for the triangles)
  u32 *blk_fp = blk_f + (size_t)g.nl * K;
  const LaneId L = lane_id(g);
  const int nblk = ((g.N + 31) >> 5) << 2;                              // dot8 phase: 8-step blocks, multiple of 4
  u32 *a8fp = blk_fp + (size_t)g.nl * K;                                // [nblk] reversed-nibble words of fp (16-byte aligned)
  const int e_alloc = g.eo_len + 2;                                     // dwords per item incl. spare front entry + dummy slot
  const int e_off0 = nblk + (L.wave * g.G + L.grp) * e_alloc + 1;
  if (L.wave == 0 && L.active && L.grp == 0) {                          // key-dependent tables, once per workgroup
    u32 av[K];
    masks_f[L.sub] = load_block_masks<K>(ValTernary{f, q - 1}, g, L.sub, av);
#pragma unroll
    for (int t = 0; t < K; t++) blk_f[K * L.sub + t] = av[t];
    masks_fp[L.sub] = load_block_masks<K>(ValU8{fp}, g, L.sub, av);
#pragma unroll
    for (int t = 0; t < K; t++) blk_fp[K * L.sub + t] = av[t];
  }
  if (D8 && L.wave == 1) {                                              // A8r[I]: nibble i = fp[8I + 7 - i]
    for (int I = L.lane; I < nblk; I += 64) {
      u32 w = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) { const int j = 8 * I + 7 - i; w |= (j < g.N ? (u32)fp[j] & 0xFu : 0u) << (4 * i); }
      a8fp[I] = w;
    }
  }
  const u32 *av_f = blk_f + K * L.sub, *av_fp = blk_fp + K * L.sub;
  __syncthreads();
  const long ngroups = (B + g.G - 1) / g.G;
  const bool want_q1 = quot1 != nullptr, want_q2 = quot2 != nullptr;
  for (long grp = (long)blockIdx.x * WAVES_PER_BLOCK + L.wave; grp < ngroups; grp += (long)gridDim.x * WAVES_PER_BLOCK) {
    // Re-materialise the lane index and N every iteration: otherwise the compiler hoists all per-lane address /
    // predicate arithmetic of the glue below out of this loop and keeps ~100 registers live (and spilled) across
    // both hot loops.
    int sub = L.sub, N = g.N, e_off = e_off0;
    asm volatile("" : "+v"(sub), "+s"(N), "+v"(e_off));
    u32 *E = a8fp + e_off;
    const long item = grp * g.G + L.grp;
    const bool valid = L.active && item < B;
    const long row = (valid ? item : 0) * N;
    // ---- product 1: a = f * e mod q, window = this item's ciphertext
    u32 P[K];
    {
      const u16 *er = e + row;
#pragma unroll
      for (int t = 0; t < K; t++) {                       // clamped indices: lanes past the end load something harmless
        const int j = 2 * (K * sub + t);
        const int j0 = j < N ? j : N - 1;
        const int j1 = j + 1 < N ? j + 1 : (j + 1 == N ? 0 : N - 1);
        P[t] = (u32)er[j0] | ((u32)er[j1] << 16);
      }
    }
    build_cyclic_pairs<K>(E, g, sub, L.active, P, false, 0u);
    // remainder1 / quotient1 are stored and the centred lift (index.js:117 verbatim) is applied pair by pair; the lifted
    // message replaces P: it is the window of product 2
    const StorePlan sp = store_plan<K>(g, sub, valid);
    const long lane0 = row + 2 * K * sub;                              // this lane's first output
    shared_product_split<K, ME, true>(E, masks_f, av_f, g, sub, want_q1, q, [&](int t, u16x2 rv, u16x2 qv) {
      if (rem1) store_pair(rem1 + lane0, sp, t, rv);
      if (want_q1) store_pair(quot1 + lane0, sp, t, qv);
      const int k = 2 * (K * sub + t);
      const u32 x0 = rv.x, x1 = rv.y;
      u32 b0 = mod_small(2 * x0 > q ? x0 + 1 : x0, p), b1 = mod_small(2 * x1 > q ? x1 + 1 : x1, p);
      b0 = k < N ? b0 : 0; b1 = k + 1 < N ? b1 : 0;
      P[t] = b0 | (b1 << 16);
    });
    wave_lds_fence();                                                   // everyone is done reading E(e)
    if constexpr (D8) {
      // ---- product 2 on v_dot8: the item's buffer now holds the nibble stream of the lifted message
      int d_off = e_off - 1, l8 = L.lane & 31;                          // 16-byte aligned start of the item's buffer
      asm volatile("" : "+v"(d_off), "+v"(l8));                         // keep this phase's addresses out of product 1
      u32 *dwp = a8fp + d_off;
      build_nibble_stream<K>(dwp, N, nblk, sub, L.active, P);
      const int lanes8 = (N + 31) >> 5;
      const bool act8 = l8 < lanes8 && L.active;
      l8 = act8 ? l8 : 0;
      if (want_q2) dot8_product_mod3<true>(dwp, a8fp, N, nblk, l8, valid && act8, lanes8, value + row, quot2 + row);
      else dot8_product_mod3<false>(dwp, a8fp, N, nblk, l8, valid && act8, lanes8, value + row, value + row);
    } else {
      build_cyclic_pairs<K>(E, g, sub, L.active, P, true, P[0] & 0xFFFFu);
      // ---- product 2: c = fp * b mod p (exact small sums, no masking)
      shared_product_split<K, 0, false>(E, masks_fp, av_fp, g, sub, want_q2, p, [&](int t, u16x2 rv, u16x2 qv) {
        store_pair(value + lane0, sp, t, rv);
        if (want_q2) store_pair(quot2 + lane0, sp, t, qv);
      });
    }
    wave_lds_fence();
  }
}

// Does any active lane of this lane's item have `pred` set?  (items occupy nl consecutive lanes of the wave)
static __device__ __forceinline__ bool item_any(bool pred, const Geom &g, const LaneId &L) {
  unsigned long long bal = __ballot(pred && L.active);
  unsigned long long msk = (g.nl >= 64 ? ~0ull : ((1ull << g.nl) - 1)) << (L.grp * g.nl);
  return (bal & msk) != 0;
}

// verifyKeysInputs, index.js:141-197, per-item key material; three products per item.
template <int K>
__global__ __launch_bounds__(BLOCK_THREADS) void k_verify_keys(
    Geom g, u32 q, u32 p, const int8_t *__restrict__ f, const int8_t *__restrict__ gg, const u16 *__restrict__ fq,
    const uint8_t *__restrict__ fp, const u16 *__restrict__ h, long B, u16 *__restrict__ quot_fq,
    u16 *__restrict__ rem_fq, uint8_t *__restrict__ quot_fp, uint8_t *__restrict__ rem_fp, u16 *__restrict__ quot_h,
    u16 *__restrict__ rem_h, uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const LaneId L = lane_id(g);
  const size_t raw_len = ((size_t)g.N + 1) & ~(size_t)1;
  const size_t per_wave = (size_t)g.G * ((size_t)g.eo_len * 8 + (size_t)g.a_len * 4 + raw_len * 2);
  unsigned char *wbase = lds + (size_t)L.wave * per_wave;
  uint2 *eo = (uint2 *)wbase + (size_t)L.grp * g.eo_len;
  u32 *a32 = (u32 *)(wbase + (size_t)g.G * g.eo_len * 8) + (size_t)L.grp * g.a_len;
  u16 *raw = (u16 *)(wbase + (size_t)g.G * ((size_t)g.eo_len * 8 + (size_t)g.a_len * 4)) + (size_t)L.grp * raw_len;
  const long ngroups = (B + g.G - 1) / g.G;
  for (long grp = (long)blockIdx.x * WAVES_PER_BLOCK + L.wave; grp < ngroups; grp += (long)gridDim.x * WAVES_PER_BLOCK) {
    const long item = grp * g.G + L.grp;
    const bool valid = L.active && item < B;
    const long row = (valid ? item : 0) * g.N;
    u32 fl = 0;
    u16x2 r[K], qv[K];
    // ---- fq * f mod q (index.js:158-160)
    stage_raw(raw, g.N, g.nl, ValTernary{f + row, q - 1}, L.sub, L.active);
    stage_a((u16 *)a32, g, ValU16{fq + row}, L.sub, L.active);
    wave_lds_fence();
    if (L.active) build_eo(eo, g, ValLds{raw}, L.sub, g.nl);
    wave_lds_fence();
    product_split<K, true>(eo, a32, g, L.sub, true, q, r, qv);
    {
      bool nz_hi = false;                             // any remainder coefficient beyond index 0 non-zero?
#pragma unroll
      for (int t = 0; t < K; t++) {
        r[t] = r[t] & (u16)(q - 1);
        int k = 2 * K * L.sub + 2 * t;
        nz_hi |= (k >= 1 && k < g.N && r[t].x != 0) || (k + 1 < g.N && r[t].y != 0);
      }
      bool first_not_one = item_any(L.sub == 0 && r[0].x != 1, g, L);
      if (item_any(nz_hi, g, L) && first_not_one) fl |= NTRU_FLAG_INVALID_FQ;   // length !== 1 && [0] !== 1
    }
    if (valid) { store_pairs<K>(rem_fq + row, g, L.sub, r); store_pairs<K>(quot_fq + row, g, L.sub, qv); }
    wave_lds_fence();
    // ---- fp * f mod p (index.js:161-163)
    // the staged f only differs in what -1 maps to: patch q-1 -> p-1 in place instead of re-reading it
    if (q > 2) {
      if (L.active) for (int i = L.sub; i < g.N; i += g.nl) { const u16 v = raw[i]; raw[i] = v == (u16)(q - 1) ? (u16)(p - 1) : v; }
    } else {                                              // q = 2: -1 and +1 coincide mod q, so re-stage from the source
      stage_raw(raw, g.N, g.nl, ValTernary{f + row, p - 1}, L.sub, L.active);
    }
    stage_a((u16 *)a32, g, ValU8{fp + row}, L.sub, L.active);
    wave_lds_fence();
    if (L.active) build_eo(eo, g, ValLds{raw}, L.sub, g.nl);
    wave_lds_fence();
    product_split<K, false>(eo, a32, g, L.sub, true, p, r, qv);
    {
      bool nz_hi = false;
#pragma unroll
      for (int t = 0; t < K; t++) {
        int k = 2 * K * L.sub + 2 * t;
        nz_hi |= (k >= 1 && k < g.N && r[t].x != 0) || (k + 1 < g.N && r[t].y != 0);
      }
      bool first_not_one = item_any(L.sub == 0 && r[0].x != 1, g, L);
      if (item_any(nz_hi, g, L) && first_not_one) fl |= NTRU_FLAG_INVALID_FP;
    }
    if (valid) { store_pairs<K>(rem_fp + row, g, L.sub, r); store_pairs<K>(quot_fp + row, g, L.sub, qv); }
    wave_lds_fence();
    // ---- (p*fq) * g mod q (index.js:155,164-166)
    stage_raw(raw, g.N, g.nl, ValTernary{gg + row, q - 1}, L.sub, L.active);
    stage_a((u16 *)a32, g, ValU16x3{fq + row, p}, L.sub, L.active);
    wave_lds_fence();
    if (L.active) build_eo(eo, g, ValLds{raw}, L.sub, g.nl);
    wave_lds_fence();
    product_split<K, true>(eo, a32, g, L.sub, true, q, r, qv);
    {
      // 'invalid h' iff some index below h's trimmed length differs from the remainder (index.js:165)
      int top = -1; bool differs_any[2 * K];
#pragma unroll
      for (int t = 0; t < K; t++) {
        r[t] = r[t] & (u16)(q - 1);
        int k = 2 * K * L.sub + 2 * t;
        u32 h0 = k < g.N ? h[row + k] : 0, h1 = k + 1 < g.N ? h[row + k + 1] : 0;
        if (h0) top = k;
        if (h1) top = k + 1;
        differs_any[2 * t] = k < g.N && h0 != r[t].x;
        differs_any[2 * t + 1] = k + 1 < g.N && h1 != r[t].y;
      }
      // degree of h over the item's lanes: highest lane holding a non-zero coefficient wins
      unsigned long long bal = __ballot(top >= 0 && L.active);
      unsigned long long msk = (g.nl >= 64 ? ~0ull : ((1ull << g.nl) - 1)) << (L.grp * g.nl);
      bal &= msk;
      int hl = 1;                                       // trimmed length of the zero polynomial is 1
      int src = bal ? 63 - __builtin_clzll(bal) : (int)L.lane;
      int top_src = __shfl(top, src);
      if (bal) hl = top_src + 1;
      bool bad = false;
#pragma unroll
      for (int j = 0; j < 2 * K; j++) bad |= differs_any[j] && (2 * K * L.sub + j) < hl;
      if (item_any(bad, g, L)) fl |= NTRU_FLAG_INVALID_H;
    }
    if (valid) {
      store_pairs<K>(rem_h + row, g, L.sub, r); store_pairs<K>(quot_h + row, g, L.sub, qv);
      if (L.sub == 0) flags[item] = (uint8_t)fl;
    }
    wave_lds_fence();
  }
}

// dividePolynomials(a, I, mod) for reduced dividends, elementwise (HBM-bound): a is [B][2N].
__global__ void k_split_by_I(int N, u32 mod, const u16 *__restrict__ a, long B, u16 *__restrict__ quot,
                             u16 *__restrict__ rem) {
  const long total = B * N;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / N; const int k = (int)(idx - row * N);
    const u32 lo = a[row * 2 * N + k], hi = a[row * 2 * N + N + k];
    quot[idx] = (u16)((mod - hi % mod) % mod);
    rem[idx] = (u16)((lo + hi) % mod);
  }
}

// addPolynomials(a, b, mod) on [B][N] rows, elementwise (HBM-bound).  The rows are contiguous, so the batch is one flat
// array: 16 bytes (8 coefficients) per lane per access when the three base pointers are 16-byte aligned.
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
template <bool POW2>
__global__ void k_add_mod_vec(u32 mod, const u16x8 *__restrict__ a, const u16x8 *__restrict__ b, long nvec,
                              u16x8 *__restrict__ out) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < nvec; idx += (long)gridDim.x * blockDim.x) {
    const u16x8 x = a[idx], y = b[idx];
    u16x8 r;
    if (POW2) {
      r = (x + y) & (u16)(mod - 1);                     // q | 2^16: wrapped 16-bit sums are exact mod q
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = (u16)(((u32)x[k] + (u32)y[k]) % mod);
    }
    out[idx] = r;
  }
}
__global__ void k_add_mod(u32 mod, const u16 *__restrict__ a, const u16 *__restrict__ b, long first, long total,
                          u16 *__restrict__ out) {
  for (long idx = first + (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
    out[idx] = (u16)(((u32)a[idx] + (u32)b[idx]) % mod);
}


// verifyKeysInputs on the add path: every product steps over a per-item ternary operand (f, f, g) with a per-item
// window (fq, fp, 3*fq mod q), one item per wave.  Needs p == 3 (so that -1 = 2 mod p is the "other" symbol).
template <int K, int ME>
__global__ __launch_bounds__(BLOCK_THREADS) void k_verify_keys_t(
    Geom g, u32 q, u32 p, const int8_t *__restrict__ f, const int8_t *__restrict__ gg, const u16 *__restrict__ fq,
    const uint8_t *__restrict__ fp, const u16 *__restrict__ h, long B, u16 *__restrict__ quot_fq,
    u16 *__restrict__ rem_fq, uint8_t *__restrict__ quot_fp, uint8_t *__restrict__ rem_fp, u16 *__restrict__ quot_h,
    u16 *__restrict__ rem_h, uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const LaneId L = lane_id(g);
  const size_t raw_len = ((size_t)g.N + 1) & ~(size_t)1;
  const size_t per_wave = (size_t)g.eo_len * 8 + (size_t)g.nl * 4 + raw_len * 2;
  unsigned char *wbase = lds + (size_t)L.wave * per_wave;
  uint2 *eo = (uint2 *)wbase;
  u32 *codes = (u32 *)(wbase + (size_t)g.eo_len * 8);
  u16 *raw = (u16 *)(wbase + (size_t)g.eo_len * 8 + (size_t)g.nl * 4);
  for (long item = (long)blockIdx.x * WAVES_PER_BLOCK + L.wave; item < B; item += (long)gridDim.x * WAVES_PER_BLOCK) {
    int sub = L.sub, N = g.N;
    asm volatile("" : "+v"(sub), "+s"(N));
    const long row = item * N;
    u32 fl = 0;
    // one product: window `win` (already mapped into [0, mod)), stepping operand `step` (ternary, -1 -> neg)
    auto product = [&](auto win, const int8_t *step, u32 neg, auto split, auto emit) {
      asm volatile("" : "+v"(sub));                      // fresh lane index: no load of this product is CSE'd with another's
      stage_raw(raw, N, g.nl, win, sub, L.active);
      {
        u32 av[K];
        const u32 word = load_block<K>(ValTernary{step + row, neg}, N, sub, av);
        if (L.active) codes[sub] = word;
      }
      wave_lds_fence();
      if (L.active) build_eo(eo, g, ValLds{raw}, sub, g.nl);
      wave_lds_fence();
      auto s_pair = [&](int t) -> u32 {                                   // reloaded behind the hot loop
        int sub2 = sub;
        asm volatile("" : "+v"(sub2));
        const int8_t *sr = step + row;
        const int k = 2 * K * sub2 + 2 * t;
        const int k0 = k < N ? k : N - 1, k1 = k + 1 < N ? k + 1 : N - 1;
        const int a0 = sr[k0], a1 = sr[k1];
        const u32 v0 = k < N ? (a0 < 0 ? neg : (u32)a0) : 0u, v1 = k + 1 < N ? (a1 < 0 ? neg : (u32)a1) : 0u;
        return v0 | (v1 << 16);
      };
      split(s_pair, emit);
      wave_lds_fence();
    };
    // per-product bookkeeping shared by the two inverse checks (index.js:159,162): "length != 1 && [0] != 1"
    bool nz_hi = false, first_not_one = false;
    auto note_inverse = [&](int sub2, int t, u16x2 rv) {
      const int k = 2 * K * sub2 + 2 * t;
      nz_hi |= (k >= 1 && k < N && rv.x != 0) || (k + 1 < N && rv.y != 0);
      if (t == 0) first_not_one = sub2 == 0 && rv.x != 1;
    };
    // ---- fq * f mod q
    product(ValU16{fq + row}, f, q - 1,
            [&](auto s_pair, auto emit) { tern_product_split<K, ME, true>(eo, codes, s_pair, g, sub, true, q, emit); },
            [&](int t, u16x2 rv, u16x2 qv) {
              int sub2 = sub; asm volatile("" : "+v"(sub2));
              rv = rv & (u16)(q - 1);
              const StorePlan sp = store_plan<K>(g, sub2, L.active);
              const long lane0 = row + 2 * K * sub2;
              store_pair(rem_fq + lane0, sp, t, rv);
              store_pair(quot_fq + lane0, sp, t, qv);
              note_inverse(sub2, t, rv);
            });
    if (item_any(nz_hi, g, L) && item_any(first_not_one, g, L)) fl |= NTRU_FLAG_INVALID_FQ;
    // ---- fp * f mod p
    nz_hi = false; first_not_one = false;
    product(ValU8{fp + row}, f, p - 1,
            [&](auto s_pair, auto emit) { tern_product_split<K, 0, false>(eo, codes, s_pair, g, sub, true, p, emit); },
            [&](int t, u16x2 rv, u16x2 qv) {
              int sub2 = sub; asm volatile("" : "+v"(sub2));
              const StorePlan sp = store_plan<K>(g, sub2, L.active);
              const long lane0 = row + 2 * K * sub2;
              store_pair(rem_fp + lane0, sp, t, rv);
              store_pair(quot_fp + lane0, sp, t, qv);
              note_inverse(sub2, t, rv);
            });
    if (item_any(nz_hi, g, L) && item_any(first_not_one, g, L)) fl |= NTRU_FLAG_INVALID_FP;
    // ---- (p*fq) * g mod q: the window is reduced mod q (same product; the unreduced p*fq of index.js:155 is host glue)
    // 'invalid h' iff some index below h's trimmed length differs from the remainder (index.js:165): get that length first
    int hl = 1;
    {
      int top = -1;
#pragma unroll
      for (int t = 0; t < K; t++) {
        const int k = 2 * K * sub + 2 * t;
        const int k0 = k < N ? k : N - 1, k1 = k + 1 < N ? k + 1 : N - 1;
        if (k < N && h[row + k0]) top = k;
        if (k + 1 < N && h[row + k1]) top = k + 1;
      }
      const unsigned long long bal = __ballot(top >= 0 && L.active);
      const int src = bal ? 63 - __builtin_clzll(bal) : (int)L.lane;
      const int top_src = __shfl(top, src);
      if (bal) hl = top_src + 1;
    }
    bool bad = false;
    product(ValU16x3m{fq + row, p, q - 1}, gg, q - 1,
            [&](auto s_pair, auto emit) { tern_product_split<K, ME, true>(eo, codes, s_pair, g, sub, true, q, emit); },
            [&](int t, u16x2 rv, u16x2 qv) {
              int sub2 = sub; asm volatile("" : "+v"(sub2));
              rv = rv & (u16)(q - 1);
              const StorePlan sp = store_plan<K>(g, sub2, L.active);
              const long lane0 = row + 2 * K * sub2;
              store_pair(rem_h + lane0, sp, t, rv);
              store_pair(quot_h + lane0, sp, t, qv);
              const int k = 2 * K * sub2 + 2 * t;
              const int k0 = k < N ? k : N - 1, k1 = k + 1 < N ? k + 1 : N - 1;
              bad |= (k < hl && h[row + k0] != rv.x) || (k + 1 < hl && h[row + k1] != rv.y);
            });
    if (item_any(bad, g, L)) fl |= NTRU_FLAG_INVALID_H;
    if (L.active && sub == 0) flags[item] = (uint8_t)fl;
  }
}


// ---- on-device ternary sampler: generateCustomArray (index.js:461-488) for one item per LANE ---------------------
// Same procedure as the reference: [1]*n1 ++ [other]*n2 ++ [0]*..., then for i = N-1 .. 1: j = u32 % (i+1), swap.
// The u32 of step t of item b is word t of the ChaCha20 keystream (RFC 8439 block function) under the caller's key with
// nonce (b_lo, b_hi, "NTRU"), so any host can replay it with a stock ChaCha20.  The Fisher-Yates chain is inherently
// sequential per item, so items are spread over lanes.  A lane's row lives in LDS as 2-bit symbols (0, 1, 2 = `other`),
// 16 per dword, pitch = odd number of dwords (the lock-step accesses of all lanes hit distinct banks): 13 KB per wave
// instead of 52 KB as bytes, i.e. 12 waves per CU instead of 3.  i is wave-uniform, so u32 % (i+1) is a multiply by a
// per-step reciprocal from an LDS table (floor(2^32 / d), one correction) instead of a 35-instruction division.
struct ChaChaKey { u32 k[8]; };

#define CHACHA_QR(a, b, c, d)                                                          \
  a += b; d ^= a; d = __builtin_rotateleft32(d, 16); c += d; b ^= c; b = __builtin_rotateleft32(b, 12); \
  a += b; d ^= a; d = __builtin_rotateleft32(d, 8);  c += d; b ^= c; b = __builtin_rotateleft32(b, 7);

__global__ __launch_bounds__(64) void k_sample_ternary(int N, int n1, int n2, u32 other, ChaChaKey key,
                                                       unsigned long long first_item, long B,
                                                       uint8_t *__restrict__ out, int pd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u32 *recip = (u32 *)lds;                               // [N + 1]: floor(2^32 / d)
  u32 *rows = recip + ((N + 2) & ~1);                    // [64][pd] dwords of 16 symbols
  const int lane = threadIdx.x;
  for (int d = lane; d <= N; d += 64) recip[d] = d >= 2 ? (u32)(0x100000000ULL / (unsigned)d) : 0u;
  for (long base = (long)blockIdx.x * 64; base < B; base += (long)gridDim.x * 64) {
    // all 64 rows start identical: fill them cooperatively, one dword at a time
    for (int idx = lane; idx < 64 * pd; idx += 64) {
      const int c = (idx % pd) * 16;
      u32 w = 0;
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const int k = c + b;
        w |= (k < n1 ? 1u : (k < n1 + n2 ? 2u : 0u)) << (2 * b);
      }
      rows[idx] = w;
    }
    wave_lds_fence();
    u32 *row = rows + (size_t)lane * pd;
    const unsigned long long item = first_item + (unsigned long long)(base + lane);
    const u32 n0 = (u32)item, nn1 = (u32)(item >> 32), nn2 = 0x4e545255u;
    int i = N - 1;
    for (u32 ctr = 0; i >= 1; ctr++) {                  // i is the same in every lane: uniform loop
      u32 x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
      u32 x4 = key.k[0], x5 = key.k[1], x6 = key.k[2], x7 = key.k[3], x8 = key.k[4], x9 = key.k[5], x10 = key.k[6],
          x11 = key.k[7], x12 = ctr, x13 = n0, x14 = nn1, x15 = nn2;
      for (int r = 0; r < 10; r++) {
        CHACHA_QR(x0, x4, x8, x12) CHACHA_QR(x1, x5, x9, x13) CHACHA_QR(x2, x6, x10, x14) CHACHA_QR(x3, x7, x11, x15)
        CHACHA_QR(x0, x5, x10, x15) CHACHA_QR(x1, x6, x11, x12) CHACHA_QR(x2, x7, x8, x13) CHACHA_QR(x3, x4, x9, x14)
      }
      const u32 ks[16] = {x0 + 0x61707865u, x1 + 0x3320646eu, x2 + 0x79622d32u, x3 + 0x6b206574u,
                          x4 + key.k[0], x5 + key.k[1], x6 + key.k[2], x7 + key.k[3], x8 + key.k[4], x9 + key.k[5],
                          x10 + key.k[6], x11 + key.k[7], x12 + ctr, x13 + n0, x14 + nn1, x15 + nn2};
#pragma unroll
      for (int w = 0; w < 16; w++) {
        if (i >= 1) {
          const u32 d = (u32)(i + 1);
          u32 j = ks[w] - __umulhi(ks[w], recip[d]) * d;               // in [0, 2d)
          j = j >= d ? j - d : j;
          const int wi = i >> 4, si = 2 * (i & 15), wj = (int)(j >> 4), sj = 2 * (int)(j & 15);
          const u32 a = row[wi], b = row[wj];
          const u32 x = ((a >> si) ^ (b >> sj)) & 3u;                    // swap two 2-bit fields by their difference
          const u32 na = a ^ (x << si);
          row[wi] = na;
          row[wj] = (wi == wj ? na : b) ^ (x << sj);                     // same dword: the second store wins
          i--;
        }
      }
    }
    wave_lds_fence();
    // rows -> row-major byte output, coalesced: the wave walks one row at a time
    for (int rr = 0; rr < 64; rr++) {
      if (base + rr >= B) break;
      uint8_t *dst = out + (size_t)(base + rr) * N;
      const u32 *src = rows + (size_t)rr * pd;
      for (int k = lane; k < N; k += 64) {
        const u32 sym = (src[k >> 4] >> (2 * (k & 15))) & 3u;
        dst[k] = (uint8_t)(sym == 2u ? other : sym);
      }
    }
    wave_lds_fence();
  }
}


// ---- key inversion: polyInv / loadPrivateKeyF (index.js:30-49, 491-514) for one key per LANE --------
// f^-1 in Z_P[x]/(x^N - 1), P = 2 or 3, by 2N - 1 Bernstein-Yang division steps on the reversed polynomials: the control
// flow is the same for every key (a conditional swap and two multiply-accumulates by per-lane scalars per step), which is
// what 64 keys in lock-step need; the reference's Euclidean algorithm returns the same polynomial because the inverse is
// unique.  f is not a unit iff the gcd left in `ff` is not a constant: reported in `flags` (the reference's own `&&`
// checks accept some non-units and return garbage for them: tests/golden/keygen_cases.json).  Polynomials are bit
// planes in LDS, [array][word][lane]; GF(3) uses two planes per polynomial (plane 0: coefficient == 1, plane 1: == 2).

// NWC > 0: the planes are NWC words per polynomial held in REGISTERS (every word loop is unrolled, so all indices are
// compile-time): no LDS traffic and 8 waves per CU instead of 3; NWC = 0: any N, planes in LDS.
template <int P, int NWC>
__global__ __launch_bounds__(64) void k_invert_key(int N, const int8_t *__restrict__ f, long B, u16 *__restrict__ out16,
                                                   uint8_t *__restrict__ out8, uint8_t *__restrict__ flags, u32 flag_bit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int PL = P == 2 ? 1 : 2;                      // bit planes per polynomial
  constexpr int UNR = NWC ? 64 : 1;                       // word loops: fully unrolled for register planes
  const int lane = threadIdx.x;
  const int NW = NWC ? NWC : (N + 1 + 31) >> 5;            // N + 1 coefficients: the reversed modulus has degree N
  u32 *base = (u32 *)lds;
  u32 regs[NWC ? 4 * PL * NWC : 1];
  auto at = [&](int arr, int pl, int w) -> u32 & {
    if constexpr (NWC > 0) return regs[(arr * PL + pl) * NWC + w];
    else return base[((arr * PL + pl) * NW + w) * 64 + lane];
  };
  enum { AF = 0, AG = 1, AV = 2, AW = 3 };
  for (long k0 = (long)blockIdx.x * 64; k0 < B; k0 += (long)gridDim.x * 64) {
    const long key = k0 + lane;
    const bool have = key < B;
    const int8_t *fk = f + (have ? key : 0) * N;
    // ff = rev(x^N - 1) = 1 - x^N, gg = rev_{N-1}(f), vv = 0, ww = 1
#pragma unroll UNR
    for (int w = 0; w < NW; w++) {
      u32 g1 = 0, g2 = 0;
      for (int b = 0; b < 32; b++) {
        const int i = 32 * w + b;                         // coefficient i of gg is f[N-1-i]
        if (i < N && have) {
          int c = fk[N - 1 - i];
          c = c < 0 ? c + P : c;
          c %= P;
          g1 |= (u32)(c == 1) << b;
          g2 |= (u32)(c == 2) << b;
        }
      }
      const u32 top = (32 * w <= N && N < 32 * w + 32) ? 1u << (N & 31) : 0u;   // coefficient N of ff is -1
      if (P == 2) {
        at(AF, 0, w) = (w == 0 ? 1u : 0u) | top; at(AG, 0, w) = g1;
        at(AV, 0, w) = 0; at(AW, 0, w) = w == 0 ? 1u : 0u;
      } else {
        at(AF, 0, w) = w == 0 ? 1u : 0u; at(AF, 1, w) = top;
        at(AG, 0, w) = g1; at(AG, 1, w) = g2;
        at(AV, 0, w) = 0; at(AV, 1, w) = 0; at(AW, 0, w) = w == 0 ? 1u : 0u; at(AW, 1, w) = 0;
      }
    }
    int delta = 1;
    // GF(3) helpers on (is-one, is-two) plane pairs
    auto add3 = [](u32 a0, u32 a1, u32 b0, u32 b1, u32 &r0, u32 &r1) {
      const u32 az = ~(a0 | a1), bz = ~(b0 | b1);
      r0 = (a0 & bz) | (az & b0) | (a1 & b1);
      r1 = (a1 & bz) | (az & b1) | (a0 & b0);
    };
    for (int step = 0; step < 2 * N - 1; step++) {
      const u32 f0w0 = at(AF, 0, 0), g0w0 = at(AG, 0, 0);
      const u32 f0w1 = P == 3 ? at(AF, 1, 0) : 0u, g0w1 = P == 3 ? at(AG, 1, 0) : 0u;
      const int fc = (int)(f0w0 & 1u) + 2 * (int)(f0w1 & 1u), gc = (int)(g0w0 & 1u) + 2 * (int)(g0w1 & 1u);   // constant terms
      const bool swap = delta > 0 && gc != 0;
      const u32 sm = swap ? ~0u : 0u;
      delta = (swap ? -delta : delta) + 1;
      const int c1 = swap ? gc : fc;                      // new f(0): multiplies g and w
      const int c2 = (P - (swap ? fc : gc)) % P;           // -(new g(0)): multiplies f and v
      const u32 c2m1 = c2 == 1 ? ~0u : 0u;
      // GF(3): g and w are scaled by the unit 1 / f(0) every step (they stay consistent with each other, and the inverse
      // is unique), so ONE scalar multiplies f and v: new g = (g - (g(0) / f(0)) f) / x, and 1 / f(0) = f(0) in GF(3)
      const int cm = P == 3 ? (9 - (swap ? fc : gc) * c1) % 3 : 0;
      const bool k1 = cm == 1, k2 = cm == 2;
      u32 vcar[PL], gprev[PL];
#pragma unroll
      for (int pl = 0; pl < PL; pl++) { vcar[pl] = 0; gprev[pl] = 0; }
#pragma unroll UNR
      for (int w = 0; w < NW; w++) {
        u32 F[PL], G[PL], V[PL], W[PL];
#pragma unroll
        for (int pl = 0; pl < PL; pl++) {
          F[pl] = at(AF, pl, w); G[pl] = at(AG, pl, w); W[pl] = at(AW, pl, w);
          const u32 v = at(AV, pl, w);
          V[pl] = (v << 1) | vcar[pl];                    // v = x v
          vcar[pl] = v >> 31;
          if (P == 2) {
            u32 t = sm & (F[pl] ^ G[pl]); F[pl] ^= t; G[pl] ^= t;    // conditional swaps
            t = sm & (V[pl] ^ W[pl]); V[pl] ^= t; W[pl] ^= t;
          } else {                                                    // ... as selects: the condition is per lane, not per bit
            const u32 f_ = F[pl], v_ = V[pl];
            F[pl] = swap ? G[pl] : f_; G[pl] = swap ? f_ : G[pl];
            V[pl] = swap ? W[pl] : v_; W[pl] = swap ? v_ : W[pl];
          }
        }
        u32 NG[PL], NWW[PL];
        if (P == 2) {                                     // c1 = 1; c2 = g(0)
          NG[0] = G[0] ^ (c2m1 & F[0]);
          NWW[0] = W[0] ^ (c2m1 & V[0]);
        } else {
          const u32 b0 = k1 ? F[0] : (k2 ? F[1] : 0u), b1 = k1 ? F[1] : (k2 ? F[0] : 0u);      // cm * f
          add3(G[0], G[1], b0, b1, NG[0], NG[1]);
          const u32 d0 = k1 ? V[0] : (k2 ? V[1] : 0u), d1 = k1 ? V[1] : (k2 ? V[0] : 0u);      // cm * v
          add3(W[0], W[1], d0, d1, NWW[0], NWW[1]);
        }
#pragma unroll
        for (int pl = 0; pl < PL; pl++) {
          at(AF, pl, w) = F[pl]; at(AV, pl, w) = V[pl]; at(AW, pl, w) = NWW[pl];
          if (w > 0) at(AG, pl, w - 1) = (gprev[pl] >> 1) | (NG[pl] << 31);      // g = g / x, one word behind
          gprev[pl] = NG[pl];
        }
      }
#pragma unroll
      for (int pl = 0; pl < PL; pl++) at(AG, pl, NW - 1) = gprev[pl] >> 1;
    }
    // unit iff the gcd (in ff) is a non-zero constant
    u32 rest = 0;
#pragma unroll UNR
    for (int w = 0; w < NW; w++)
#pragma unroll
      for (int pl = 0; pl < PL; pl++) rest |= at(AF, pl, w) & (w == 0 ? ~1u : ~0u);
    const int fc = (int)(at(AF, 0, 0) & 1u) + (P == 3 ? 2 * (int)(at(AF, 1, 0) & 1u) : 0);
    const bool unit = rest == 0 && fc != 0;
    if (have) {
      if (!unit) flags[key] = (uint8_t)(flags[key] | flag_bit);
      // inverse[i] = fc^-1 * vv[N-1-i]; in GF(3) fc^-1 = fc
#pragma unroll UNR
      for (int w = 0; w < NW; w++) {
        const u32 p0 = at(AV, 0, w), p1 = P == 3 ? at(AV, 1, w) : 0u;
        for (int b = 0; b < 32; b++) {
          const int i = N - 1 - (32 * w + b);
          if (i < 0) break;
          int c = (int)((p0 >> b) & 1u) + 2 * (int)((p1 >> b) & 1u);
          if (P == 3 && fc == 2) c = (2 * c) % 3;
          c = unit ? c : 0;
          if (out16) out16[key * N + i] = (u16)c;
          if (out8) out8[key * N + i] = (uint8_t)c;
        }
      }
    }
  }
}

// f in {-1,0,1} -> its residue mod q as u16 (elementwise)
__global__ void k_signed_to_u16(const int8_t *__restrict__ f, long n, u32 q, u16 *__restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int v = f[i];
    out[i] = (u16)(v < 0 ? (u32)(v + (int)q) : (u32)v);
  }
}
// one Newton round of polyInv (index.js:499-506): v <- (2 v - u) mod q, u = f * v * v
__global__ void k_newton_combine(u16 *__restrict__ v, const u16 *__restrict__ u, long n, u32 q) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    v[i] = (u16)((2u * v[i] - u[i]) & (q - 1));
}

// ---- BN254 field-element packing (index.js:572-620): elementwise, HBM-bound ---------------------------------------
// One thread per 64-bit limb of the output: out[b][o] = sum_j data[b][o*per + j] << (j*bits), four LE limbs per element.
__global__ void k_pack(int bits, int per, int data_len, int out_size, const u16 *__restrict__ data, long B,
                       unsigned long long *__restrict__ out) {
  const long total = B * out_size * 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int l = (int)(idx & 3);
    const long eo = idx >> 2;
    const long b = eo / out_size;
    const int o = (int)(eo - b * out_size);
    const int lo_bit = 64 * l;
    int j0 = lo_bit / bits, j1 = (lo_bit + 63) / bits;
    if (j1 >= per) j1 = per - 1;
    unsigned long long limb = 0;
    for (int j = j0; j <= j1; j++) {
      const int i = o * per + j;
      const unsigned long long v = i < data_len ? data[b * data_len + i] : 0ull;
      const int sh = j * bits - lo_bit;
      limb |= sh >= 0 ? v << sh : v >> (-sh);
    }
    out[idx] = limb;
  }
}

// unpackInput before trimming: out[b][i*per + j] = (in[b][i] >> (j*bits)) & mask.
__global__ void k_unpack(int bits, int per, int packed_size, const unsigned long long *__restrict__ in, long B,
                         u16 *__restrict__ out) {
  const long total = B * packed_size * per;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long e = idx / per;
    const int j = (int)(idx - e * per);
    const unsigned long long *l = in + e * 4;
    const int pos = j * bits, w = pos >> 6, sft = pos & 63;
    unsigned long long v = l[w] >> sft;
    if (sft + bits > 64 && w + 1 < 4) v |= l[w + 1] << (64 - sft);
    out[idx] = (u16)(v & ((1u << bits) - 1u));
  }
}

// ---- family 4: shared-key products on the int8 matrix cores ---------------------------------------------------------
// A product of a batch operand X[b][i] with a SHARED key operand s is a matrix product with the Toeplitz matrix of s,
// which is what v_mfma_i32_32x32x32_i8 is for (exact: int8 x int8 -> int32).  Rows = 32 items of a row block, columns
// = output coefficients k = 32 kb + k', contraction i = 32 ib + i'; the key tile depends only on d = kb - ib:
//     G_d[i'][k'] = sc[32 d + k' - i'],  sc = s with period N
// d > 0 accumulates into `low` (coefficients 0..N-1 of the linear product), d < 0 into `high` (N..2N-1), d = 0 is split
// by k' >= i'; remainder = low + high, quotient = -high (closed form of the division by 1 - x^N).
// Operands wider than int8 use two digit planes on the contraction axis, value = lo + 128 hi computed as
// [A | alpha A] x [lo ; beta hi], alpha beta = 128.  LDS images: operand stages [32 rows][pitchA] (ds_read_b128, pitch an
// odd multiple of 16 bytes), key arrays reversed and cyclic, rev[y] = digit(sc[(32 NT - 1 - y) mod N]), in 4 byte-shifted
// copies so that a lane's 16-byte Toeplitz fragment (which starts at an arbitrary byte) is 4 aligned dwords; each of the
// 4 waves owns strips of <= 4 column tiles, its fragment window slides by one tile per contraction step (one new
// fragment per step).  tools/mfma_model.py is the executable specification; profiles/r01_microbench_mfma_lds.txt holds
// the measurements behind the layout choices.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v2u __attribute__((__vector_size__(2 * sizeof(unsigned int))));
#ifndef ST_AUX
#define ST_AUX 0     // cache policy bits of the result stores (0 measured best; 2 = non-temporal is 1.6x slower)
#endif

struct MGeom {
  int N;        // ring size
  int NT;       // 32-wide tiles per row: ceil(N / 32)
  int pitchA;   // bytes per row of an operand stage: 32 NT + 16
  int tpitch;   // dwords per byte-shifted copy of a reversed key array (= 8 mod 32: the 4 copies use disjoint banks)
  int ld;       // row pitch of every batch array in ELEMENTS (>= N; N for the dense layout of the plain entry points)
};

enum { M_ENC = 0, M_DEC1 = 1, M_DEC2 = 2 };

// -DNTRU_STAMPS: diagnostic build that records s_memtime at the phase boundaries of the matrix-core kernels for the
// first row blocks of each workgroup (tools/phase_stamps.py reads them back); no stamp executes in the shipped library.
#ifdef NTRU_STAMPS
#define STAMP_SLOTS 24
#define STAMP_BLOCKS 6
__device__ unsigned long long g_stamps[1024][8][STAMP_BLOCKS][STAMP_SLOTS];     // [workgroup][wave: 8 in the lock-step kernels]
#define STAMP(slot)                                                                                          \
  do {                                                                                                       \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024 && stamp_iter < STAMP_BLOCKS)                           \
      g_stamps[blockIdx.x][threadIdx.x >> 6][stamp_iter][slot] = __builtin_amdgcn_s_memtime();               \
  } while (0)
extern "C" int ntru_debug_read_stamps(void *dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#else
#define STAMP(slot) do { } while (0)
#endif

// -DNTRU_ABLATE=1|2|3 builds timing-only variants (1: no result stores, 2: no matrix loops); never shipped.
#if defined(NTRU_ABLATE) && (NTRU_ABLATE & 1)
#define ABL_STORE(x) && (x) == 0x7fffffff
#else
#define ABL_STORE(x)
#endif

template <class D>
static __device__ __forceinline__ void build_toeplitz_array(u32 *T, const MGeom &g, D digit, int tid, int nthr) {
  const int Y0 = 32 * g.NT - 1;
  for (int x = tid; x < 4 * g.tpitch; x += nthr) {
    const int c = x / g.tpitch, w = x - c * g.tpitch;
    u32 v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int idx = (Y0 - (4 * w + c + j)) % g.N;       // remainder takes the dividend's sign
      idx += idx < 0 ? g.N : 0;
      v |= ((u32)digit(idx) & 0xFFu) << (8 * j);
    }
    T[x] = v;
  }
}

// Per-lane pointer to the fragment of tile offset d = 0 (the fragment of offset d starts 8 d dwords below it).
static __device__ __forceinline__ const u32 *frag_lane_base(const u32 *T, const MGeom &g, int lane) {
  const int y0 = 32 * g.NT - 1 - (lane & 31) + 16 * (lane >> 5);
  return T + (y0 & 3) * g.tpitch + (y0 >> 2);
}

static __device__ __forceinline__ v4i and4(v4i a, const u32 (&m)[4]) {
  return (v4i){(int)((u32)a[0] & m[0]), (int)((u32)a[1] & m[1]), (int)((u32)a[2] & m[2]), (int)((u32)a[3] & m[3])};
}

// One strip of NT_S column tiles starting at tile kb0, all 32 rows of the staged row block.  st0 / st1: this lane's
// row of the operand stage(s) (+ 16 bytes for the upper half-wave); tb0 / tb1: this lane's fragment bases.
struct NoPause { __device__ __forceinline__ void operator()() const {} };
// pause / pause_ib: pause() is called exactly once, before the first contraction step ib >= pause_ib is touched (at a
// block boundary, so possibly a few steps early; after the loops when no such step exists).  The role-split decrypt
// kernel waits there for the operand columns that are still being produced.
template <int MODE, int NT_S, class Epi, class Pause = NoPause>
static __device__ __forceinline__ void toeplitz_strip(const unsigned char *__restrict__ st0,
                                                      const unsigned char *__restrict__ st1,
                                                      const u32 *__restrict__ tb0, const u32 *__restrict__ tb1,
                                                      const MGeom &g, int kb0, const u32 (&mlow)[4], Epi epi,
                                                      int stamp_iter = 0, int stamp_base = 0, int pause_ib = 0x7fffffff,
                                                      Pause pause = Pause()) {
#ifndef NTRU_ABLATE
#define NTRU_ABLATE 0
#endif
  constexpr bool TWO = MODE != M_DEC2;
  // The accumulators are never zeroed: the first matrix instruction of each takes the inline constant 0 as its C operand
  // (accL: contraction step 0, peeled below; accH: its own diagonal sub-step) -- 2 x 16 x NT_S moves per strip less.
  v16i accL[NT_S], accH[NT_S];
  v4i W0[NT_S], W1[NT_S];
  auto load_w = [&](int d, v4i &w0, v4i &w1) {
#if defined(NTRU_ABLATE) && (NTRU_ABLATE & 262144)       // timing only: no operand reads inside the loops (wrong values)
    if (d != kb0) { asm volatile("" : "+v"(w0), "+v"(w1)); return; }
#endif
    const u32 *p = tb0 - 8 * d;
    w0 = (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
    if (MODE == M_ENC) {
      const u32 *p1 = tb1 - 8 * d;
      w1 = (v4i){(int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
    } else if (MODE == M_DEC1) {                       // 64 f from f in {-1,0,1} (0xFF, 0, 1): the two low bits of every
#pragma unroll                                         // byte land in its bits 6-7; what the byte below shifts in is masked off
      for (int c = 0; c < 4; c++) w1[c] = (int)(((u32)w0[c] << 6) & 0xC0C0C0C0u);
    } else {
      w1 = w0;
    }
  };
  auto load_a = [&](int ib, v4i &a0, v4i &a1) {
#if defined(NTRU_ABLATE) && (NTRU_ABLATE & 262144)
    if (ib != 0) { asm volatile("" : "+v"(a0), "+v"(a1)); return; }
#endif
    a0 = *(const v4i *)(st0 + 32 * ib);
    if (MODE == M_ENC) {                               // 32 r: r <= 3, no carry between bytes
#pragma unroll
      for (int c = 0; c < 4; c++) a1[c] = (int)((u32)a0[c] << 5);
    } else if (MODE == M_DEC1) {
      a1 = *(const v4i *)(st1 + 32 * ib);
    } else {
      a1 = a0;
    }
  };
  auto mm = [&](v16i &acc, v4i a0, v4i a1, v4i w0, v4i w1) {
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, w0, acc, 0, 0, 0);
    if (TWO) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, w1, acc, 0, 0, 0);
  };
  auto mm_first = [&](v16i &acc, v4i a0, v4i a1, v4i w0, v4i w1) {       // first touch of an accumulator
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, w0, zero, 0, 0, 0);
    if (TWO) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, w1, acc, 0, 0, 0);
  };
  // Fragment window: at a step boundary slot t holds the fragment of tile t ("canonical").  A block of NT_S steps
  // rotates through the slots with compile-time indices (no register moves) and ends canonical again: at sub-step u tile t
  // uses slot (t - u) mod NT_S, and the fragment needed next replaces the one tile NT_S-1 just used.  Left-over steps
  // slide the window physically.  The next operand fragment is requested before the current step's products.
#pragma unroll
  for (int t = 0; t < NT_S; t++) load_w(kb0 + t, W0[t], W1[t]);
#if NTRU_ABLATE & 2
  const int kb0_ = kb0; kb0 = 0; const int NT_ = 0;
#pragma unroll
  for (int t = 0; t < NT_S; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) accH[t][i] = 0;
#else
  const int NT_ = g.NT;
#endif
  u32 mhigh[4];
#pragma unroll
  for (int c = 0; c < 4; c++) mhigh[c] = ~mlow[c];
  v4i a0, a1;
  load_a(0, a0, a1);
  // kind: 0 = all tiles low, 1 = all high, 2 = the strip's own (diagonal) steps
  auto block = [&](int ib, auto kind) {
#pragma unroll
    for (int u = 0; u < NT_S; u++) {
      v4i n0, n1;
      load_a(ib + u + 1, n0, n1);
#pragma unroll
      for (int t = 0; t < NT_S; t++) {
        constexpr int K = decltype(kind)::value;
        const int sl = (t - u + NT_S) % NT_S;
        if (K == 0 || (K == 2 && t > u)) mm(accL[t], a0, a1, W0[sl], W1[sl]);
        else if (K == 1 || (K == 2 && t < u)) mm(accH[t], a0, a1, W0[sl], W1[sl]);
        else {
          mm(accL[t], a0, a1, and4(W0[sl], mlow), and4(W1[sl], mlow));
          mm_first(accH[t], a0, a1, and4(W0[sl], mhigh), and4(W1[sl], mhigh));   // t == u: the first term of `high`
        }
      }
      load_w(kb0 - (ib + u + 1), W0[(NT_S - 1 - u) % NT_S], W1[(NT_S - 1 - u) % NT_S]);
      a0 = n0; a1 = n1;
    }
  };
  auto single = [&](int ib, v16i (&acc)[NT_S]) {
    v4i n0, n1;
    load_a(ib + 1, n0, n1);
#pragma unroll
    for (int t = 0; t < NT_S; t++) mm(acc[t], a0, a1, W0[t], W1[t]);
#pragma unroll
    for (int t = NT_S - 1; t > 0; t--) { W0[t] = W0[t - 1]; W1[t] = W1[t - 1]; }
    load_w(kb0 - (ib + 1), W0[0], W1[0]);
    a0 = n0; a1 = n1;
  };
  int ib = 0;
  bool paused = false;
  if (kb0 > 0 && NT_ > 0) {                              // contraction step 0: the first term of every `low`
    if (!std::is_same<Pause, NoPause>::value && pause_ib <= 0) { pause(); paused = true; load_a(0, a0, a1); }
    v4i n0, n1;
    load_a(1, n0, n1);
#pragma unroll
    for (int t = 0; t < NT_S; t++) mm_first(accL[t], a0, a1, W0[t], W1[t]);
#pragma unroll
    for (int t = NT_S - 1; t > 0; t--) { W0[t] = W0[t - 1]; W1[t] = W1[t - 1]; }
    load_w(kb0 - 1, W0[0], W1[0]);
    a0 = n0; a1 = n1;
    ib = 1;
  } else {                                               // the diagonal block comes first (or a timing-only build)
#pragma unroll
    for (int t = 0; t < NT_S; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) accL[t][i] = 0;
  }
  auto maybe_pause = [&](int first, int last) {          // the steps first .. last are what the next block touches
    if (!std::is_same<Pause, NoPause>::value && !paused && last >= pause_ib) {
      __builtin_amdgcn_s_setprio(0);
      pause();
      __builtin_amdgcn_s_setprio(3);
      paused = true;
      load_a(first, a0, a1);                             // it was requested before the pause: read it again
    }
  };
  __builtin_amdgcn_s_setprio(3);       // the partner wave on this SIMD is usually in a VALU phase: decrypt -3 %
  for (; ib + NT_S <= kb0; ib += NT_S) { maybe_pause(ib, ib + NT_S - 1); block(ib, std::integral_constant<int, 0>{}); }   // above the diagonal: low
  for (; ib < kb0; ib++) { maybe_pause(ib, ib); single(ib, accL); }
  maybe_pause(kb0, kb0 + NT_S - 1);
  block(kb0, std::integral_constant<int, 2>{});                                        // ib = kb0 .. kb0 + NT_S - 1
  for (ib = kb0 + NT_S; ib + NT_S <= NT_; ib += NT_S) { maybe_pause(ib, ib + NT_S - 1); block(ib, std::integral_constant<int, 1>{}); }   // below: high
  for (; ib < NT_; ib++) { maybe_pause(ib, ib); single(ib, accH); }
  __builtin_amdgcn_s_setprio(0);
  if (!std::is_same<Pause, NoPause>::value && !paused) pause();
  STAMP(stamp_base);
#if NTRU_ABLATE & 128
  if (MODE == M_DEC1) { if (accL[0][0] == 0x7fffffff) epi(accL, accH); return; }
#endif
#if NTRU_ABLATE & 256
  if (MODE == M_DEC2) { if (accL[0][0] == 0x7fffffff) epi(accL, accH); return; }
#endif
  epi(accL, accH);
  STAMP(stamp_base + 1);
#if NTRU_ABLATE & 2
  (void)kb0_;
#endif
}

// The NT column tiles are cut into 4 R strips of at most 4 tiles (sizes as even as possible, in column order); in round
// rho the four waves take the adjacent strips 4 rho .. 4 rho + 3, so neighbouring strips are stored at about the same
// time and the cache lines they share are completed in L2 instead of being written to HBM twice.  body(kb0, nt).
// The first `rem` strips are one tile wider (at N = 821: waves 0, 1 carry 7 tiles per product, waves 2, 3 six), and wave w
// of every workgroup runs on SIMD w.  The workgroups of the second half of the grid (the second resident workgroup of a CU
// under the usual dispatch order) therefore take the strips in the order 2, 3, 0, 1, so that each SIMD sees 7 + 6 tiles.
// all: the body is called for empty strips too (nt = 0) and the caller has applied the swap itself (lock-step kernels).
template <int MAXT = 4, class Body>
static __device__ __forceinline__ void for_each_strip(int NT, int wave, Body body, bool all = false) {
  const int rounds = (((NT + MAXT - 1) / MAXT) + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  const int n_str = rounds * WAVES_PER_BLOCK, base = NT / n_str, rem = NT % n_str;
  if (!all && 2 * blockIdx.x >= gridDim.x) wave ^= 2;
  for (int rho = 0; rho < rounds; rho++) {
    const int j = rho * WAVES_PER_BLOCK + wave;
    const int nt = base + (j < rem ? 1 : 0);
    if (nt > 0 || all) body(j * base + (j < rem ? j : rem), nt);
  }
}

static __device__ __forceinline__ void diag_low_mask(int lane, u32 (&mlow)[4]) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    u32 mk = 0;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) mk |= (r >= 16 * hh + 4 * c + jj) ? (0xFFu << (8 * jj)) : 0u;
    mlow[c] = mk;
  }
}

// Buffer descriptor of `bytes` bytes at p: loads beyond the end return 0 and stores beyond it are dropped, which is how
// the rows of a partial last row block are handled (the row block is rebased so that in-block offsets are small).
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void *p, long bytes) {
  // the operands are wave-uniform; saying so keeps the descriptor in scalar registers (a descriptor the compiler takes
  // for divergent turns every access into a waterfall loop)
  const unsigned long long a = (unsigned long long)p;
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)a), hi = __builtin_amdgcn_readfirstlane((u32)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7FFFF000L ? bytes : 0x7FFFF000L));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}

// Reading rows whose pitch (N or 2N bytes, N odd) is not a multiple of 16: a per-lane 16-byte load at an unaligned
// address runs at a fraction of the aligned rate (profiles/r01_ablation_mfma.txt: 0.8 of 2.5 ms), so rows are read as
// ALIGNED 16-byte chunks and shifted in registers; the shift is wave-uniform because a wave stages one row at a time.
// AlignedSrc: descriptor based at the 16-byte aligned address at or below p, a0 = p's offset in it; out-of-range
// dwords read as zero (the range check is per dword, so the size is rounded up to whole dwords).
struct AlignedSrc { __amdgpu_buffer_rsrc_t rs; int a0; };
static __device__ __forceinline__ AlignedSrc aligned_src(const void *p, long bytes) {
  const unsigned long long a = (unsigned long long)p;
  AlignedSrc s;
  s.a0 = (int)(a & 15);
  s.rs = rows_rsrc((const void *)(a & ~15ULL), (bytes + s.a0 + 3) & ~3L);
  return s;
}

// NCH consecutive 16-byte chunks starting at byte `pos` (any alignment) of src, in two steps so that a caller can put
// many loads in flight before the first shift: raw (dword-aligned loads: a 4-byte aligned 16-byte load runs at the full
// rate, only sub-dword misalignment is slow), then one v_alignbyte per dword by pos & 3.
template <int NCH>
struct RawChunks { v4i c[NCH]; u32 tail; };
template <int NCH>
static __device__ __forceinline__ RawChunks<NCH> load_raw(const AlignedSrc &src, int pos, int) {
  RawChunks<NCH> r;
  const int al = pos & ~3;
#pragma unroll
  for (int c = 0; c < NCH; c++) r.c[c] = __builtin_amdgcn_raw_buffer_load_b128(src.rs, al + 16 * c, 0, 0);
  r.tail = __builtin_amdgcn_raw_buffer_load_b32(src.rs, al + 16 * NCH, 0, 0);
  return r;
}
template <int NCH>
static __device__ __forceinline__ RawChunks<NCH> fake_raw(int v) {        // timing-only builds (NTRU_ABLATE)
  RawChunks<NCH> r;
#pragma unroll
  for (int c = 0; c < NCH; c++) r.c[c] = (v4i){v, 1, 2, 1};
  r.tail = 0;
  return r;
}
template <int NCH>
static __device__ __forceinline__ void shift_raw(const RawChunks<NCH> &r, int sh, v4i (&out)[NCH]) {
  u32 d[4 * NCH + 1];
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int k = 0; k < 4; k++) d[4 * c + k] = (u32)r.c[c][k];
  d[4 * NCH] = r.tail;
#pragma unroll
  for (int k = 0; k < 4 * NCH; k++) out[k >> 2][k & 3] = (int)__builtin_amdgcn_alignbyte(d[k + 1], d[k], (u32)(sh & 3));
}

// Byte mask of the columns < N inside the 16-byte chunk starting at column c16 (all ones / partial / zero).
static __device__ __forceinline__ v4i col_mask16(int c16, int N) {
  v4i mk;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int left = N - (c16 + 4 * k);                  // valid bytes of this dword
    mk[k] = left >= 4 ? -1 : (left <= 0 ? 0 : (int)((1u << (8 * left)) - 1u));
  }
  return mk;
}

// encryptBits on the matrix cores: e = (r * h + m) split by 1 - x^N; r in {0..3} bytes, h < q <= 8192.
// h is taken in the representative hs = d0 + 128 d1, d0 in [-64,63], 4 d1 in [-128,124]; planes [r | 32 r] x [d0 ; 4 d1].
// MAXT: widest strip.  8 (one workgroup per CU, 512 registers per wave, one strip per wave and row block) was measured at
// 2.28 ms per 2^20 against 1.51-1.58 ms for 4: with one wave per SIMD nothing overlaps the matrix loops
// (DESIGN.md section 5b); only 4 is instantiated.
// Result chunk of one wave (CHUNK variants): 8 rows x OC_PITCH bytes = a strip's <= 128 u16 columns of 8 rows, each row
// at the 16-byte phase it has in global memory (<= 14 bytes of slack in front).
constexpr int OC_PITCH = 272, OC_BYTES = 8 * OC_PITCH;

// DMA (k_encrypt_md, GROUPS = 1): the batch operands reach LDS by direct global -> LDS loads (buffer_load_dwordx4 ... lds, no
// registers; see k_decrypt_m8d).  r of the NEXT row block is requested into the r stage once every wave has left its last matrix
// loop -- in front of the last epilogue's stores instead of behind them (phase stamps: the rows requested at the top of a trip
// come back 8-9 k cycles later) -- and brought into operand form (shift to byte 0, columns >= N zeroed) in place by the wave that
// owns the row; m is requested straight into the m image at the top of a trip and only waited for before the first epilogue.
// Two more barriers per row block, all of them LDS-only.
template <int MAXT, int GROUPS, bool CHUNK = false, bool DMA = false>   // GROUPS = 2: the lock-step schedule of decrypt_m_body (k_encrypt_m8)
static __device__ __forceinline__ void encrypt_m_body(MGeom g, u32 q, const u16 *__restrict__ h,
                                                      const uint8_t *__restrict__ r,
                                                      const uint8_t *__restrict__ m, long B,
                                                      u16 *__restrict__ e, u16 *__restrict__ quotE) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  static_assert(!DMA || (GROUPS == 1 && !CHUNK), "the direct-to-LDS variant is built on the plain two-workgroup kernel");
  auto wg_barrier = [&]() {                                // DMA: LDS-only (with such loads in flight __syncthreads() waits for vmcnt(0): every store)
    if (DMA) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  };
  const int group = GROUPS == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  u32 *T0 = (u32 *)lds, *T1 = T0 + 4 * g.tpitch;         // key arrays (shared by the groups), then per group [r stage][m image]
  const int gbytes = 32 * g.pitchA + ((32 * g.ld + 15) & ~15) + 16 + (CHUNK ? WAVES_PER_BLOCK * OC_BYTES : 0);
  unsigned char *stA = (unsigned char *)(T1 + 4 * g.tpitch) + group * gbytes;
  unsigned char *mimg = stA + 32 * g.pitchA;             // rows b0..b0+31 of m exactly as in memory (pitch g.ld)
  unsigned char *ochunks = mimg + ((32 * g.ld + 15) & ~15) + 16;   // CHUNK: one result chunk per wave
  const int tid0 = threadIdx.x & (BLOCK_THREADS - 1), lane0 = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int hthr = (int)(q >> 1) - 65;
  auto hs_of = [&](int i) { int hv = (int)(h[i] & (q - 1)); return hv > hthr ? hv - (int)q : hv; };
  build_toeplitz_array(T0, g, [&](int i) { const int hs = hs_of(i); return ((hs + 64) & 127) - 64; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
  build_toeplitz_array(T1, g, [&](int i) { const int hs = hs_of(i); const int d0 = ((hs + 64) & 127) - 64; return ((hs - d0) >> 7) * 4; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
  const bool want_q = quotE != nullptr;
  const long nrb = (B + 31) >> 5;
  int sidx = 0, stamp_iter = -1;
  auto phase = [&]() { if (GROUPS == 2) wg_barrier(); };
  if (GROUPS == 2 && group == 1) wg_barrier();          // group 1 runs one phase behind group 0
  const long stride = (long)gridDim.x * GROUPS, iters = (nrb + stride - 1) / stride;
  const int rounds = (((g.NT + MAXT - 1) / MAXT) + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  // DMA: pieces start at absolutely dword-aligned addresses (descriptor based at the dword at or below the row block, size in
  // whole dwords: see k_decrypt_m8d), so a row lands 0-3 bytes into its slot and the image 0-3 bytes into the m image.
  auto dma_r = [&](long rbx, int lane) {                    // rows wave, wave + 4, ... into the r stage, one instruction per row
    const long b0x = rbx << 5 < B ? rbx << 5 : B;
    const unsigned long long a = (unsigned long long)(r + b0x * g.ld);
    const int a0 = (int)(a & 3);
    const __amdgpu_buffer_rsrc_t rs = rows_rsrc((const void *)(a & ~3ULL), ((B - b0x) * g.ld + a0 + 3) & ~3L);
#pragma unroll
    for (int j = 0; j < 32 / WAVES_PER_BLOCK; j++) {
      const int row = wave + WAVES_PER_BLOCK * j, ro = a0 + row * g.ld;
      if (lane < (((ro & 3) + g.N + 15) >> 4))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(stA + row * g.pitchA), 16,
                                                 (ro & ~3) + 16 * lane, 0, 0, 0);
    }
  };
  auto dma_m = [&](long rbx, int tid) {                     // the 32 rows of m as one run of 16-byte pieces into the m image
    const long b0x = rbx << 5 < B ? rbx << 5 : B;
    const unsigned long long a = (unsigned long long)(m + b0x * g.ld);
    const int a0 = (int)(a & 3);
    const __amdgpu_buffer_rsrc_t rs = rows_rsrc((const void *)(a & ~3ULL), ((B - b0x) * g.ld + a0 + 3) & ~3L);
    const int npc = (a0 + 32 * g.ld + 15) >> 4;
#pragma unroll
    for (int j = 0; j < 8; j++)                             // 8 x 256 pieces >= 32 x 1024 / 16
      if (tid + j * BLOCK_THREADS < npc)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(mimg + 16 * (wave * 64 + j * BLOCK_THREADS)), 16,
                                                 16 * (tid + j * BLOCK_THREADS), 0, 0, 0);
  };
  if (DMA) {
    dma_r((long)blockIdx.x < nrb ? (long)blockIdx.x : nrb, lane0);
    __builtin_amdgcn_s_waitcnt(0);                         // nothing else is in flight yet
  }
  for (long it = 0; it < iters; it++) {
    long rb = (long)blockIdx.x * GROUPS + group + it * stride;   // past the end: a row block of zeros whose stores are dropped
    rb = rb < nrb ? rb : nrb;
    long rb_next = (long)blockIdx.x * GROUPS + group + (it + 1) * stride;
    rb_next = rb_next < nrb ? rb_next : nrb;
    stamp_iter++;
    STAMP(0);
    // Re-materialise the lane index and N per row block: otherwise every per-lane address / predicate of the staging
    // and of the epilogues is hoisted out of this loop and spilled around the matrix loops.
    int lane = lane0, N = g.N, LD = g.ld, tid = tid0;
    asm volatile("" : "+v"(lane), "+s"(N), "+s"(LD), "+v"(tid));
    const u32 *tb0 = frag_lane_base(T0, g, lane), *tb1 = frag_lane_base(T1, g, lane);
    const unsigned char *st0 = stA + (lane & 31) * g.pitchA + 16 * (lane >> 5);
    u32 mlow[4];
    diag_low_mask(lane, mlow);
    const long b0 = rb << 5 < B ? rb << 5 : B, left = (B - b0) * LD;   // elements from this row block to the end of the batch
    const AlignedSrc src_r = aligned_src(r + b0 * LD, left), src_m = aligned_src(m + b0 * LD, left);
    // All loads of the row block (r rows and the m image) are requested BEFORE the barrier: they land in registers, so
    // they need not wait for the previous row block's readers, and the two HBM round trips become one that overlaps the
    // barrier wait (phase stamps: 4.5 k + 4.5 k cycles back to back before).  N <= 1024: lane = 16-byte chunk of a row.
    constexpr int RPW = 32 / WAVES_PER_BLOCK;
    const int shm = __builtin_amdgcn_readfirstlane(src_m.a0);
    RawChunks<1> in_r[RPW], in_m[8];
    if (!DMA) {
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int pos0 = src_r.a0 + (wave + WAVES_PER_BLOCK * j) * LD;
#if defined(NTRU_ABLATE) && (NTRU_ABLATE & 4)
        in_r[j] = fake_raw<1>(pos0 + lane);
#else
        in_r[j] = load_raw<1>(src_r, pos0 + 16 * lane, 0);
#endif
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i = tid * 16 + j * BLOCK_THREADS * 16;
#if defined(NTRU_ABLATE) && (NTRU_ABLATE & 8)
        in_m[j] = fake_raw<1>(i);
#else
        in_m[j] = load_raw<1>(src_m, src_m.a0 + i, 0);   // past the row block: next rows or zeros, not written
#endif
      }
    }
    wg_barrier();                                    // the previous row block's readers are done (first pass: key arrays built)
    STAMP(1);
    const int a0m = DMA ? __builtin_amdgcn_readfirstlane((int)((unsigned long long)(m + b0 * LD) & 3)) : 0;   // DMA: the m image starts a0m bytes in
    if (DMA) {                                       // every wave's r rows have landed (each waited for its own in its last epilogue)
      dma_m(rb, tid);
      const int a0r = (int)((unsigned long long)(r + b0 * LD) & 3);
      const v4i mk = col_mask16(16 * lane, N);
      int s4[RPW];
#pragma unroll
      for (int j = 0; j < RPW; j++) {                // all reads of the wave's rows before the first write back
        const int row = wave + WAVES_PER_BLOCK * j;
        const unsigned char *slot = stA + row * g.pitchA + 16 * (lane < 2 * g.NT ? lane : 0);
        in_r[j].c[0] = *(const v4i *)slot;
        in_r[j].tail = *(const u32 *)(slot + 16);
        s4[j] = __builtin_amdgcn_readfirstlane((a0r + row * LD) & 3);
      }
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int row = wave + WAVES_PER_BLOCK * j;
        v4i v[1];
        shift_raw<1>(in_r[j], s4[j], v);
        if (lane < 2 * g.NT) *(v4i *)(stA + row * g.pitchA + 16 * lane) = v[0] & mk;
      }
    } else {
      const v4i mk = col_mask16(16 * lane, N);
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int row = wave + WAVES_PER_BLOCK * j;
        v4i v[1];
        shift_raw<1>(in_r[j], src_r.a0 + row * LD, v);
        if (lane < 2 * g.NT) *(v4i *)(stA + row * g.pitchA + 16 * lane) = v[0] & mk;
      }
      STAMP(16);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i = tid * 16 + j * BLOCK_THREADS * 16;
        v4i v[1];
        shift_raw<1>(in_m[j], shm, v);
        if (i < 32 * LD) *(v4i *)(mimg + i) = v[0];
      }
    }
    STAMP(2);
    wg_barrier();
    STAMP(3);
    sidx = 0;
    // DMA: what a wave does at the start of an epilogue, strip or no strip (every wave walks through every round): wait for its m
    // pieces before the first one, barriers before the first (m image complete) and the last (r stage free of readers), then the
    // request for the next row block's r rows -- in front of this epilogue's stores.
    auto epi_sync = [&]() {
      if (!DMA) return;
      if (sidx == 0) __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));          // vmcnt(0): this wave's m pieces (the loops issue no vector memory operation)
      if (sidx == 0 || sidx == rounds - 1) wg_barrier();
      if (sidx == rounds - 1 && it + 1 < iters) dma_r(rb_next, lane);
    };
    const bool dma_now = DMA && it + 1 < iters;          // (in the last round) r rows of the next row block are in flight behind this epilogue's stores
    for_each_strip<MAXT>(g.NT, GROUPS == 2 ? wave ^ (2 * group) : (DMA ? wave ^ (2 * (int)blockIdx.x >= (int)gridDim.x ? 2 : 0) : wave), [&](int kb0, int nt) {
      // Result register i of a tile is row (i & 3) + 8 (i >> 2) + 4 (lane >> 5), column lane & 31: a per-lane offset
      // plus a wave-uniform (scalar) offset per register; rows past the batch end are dropped by the descriptor.
      // (Packing 4 columns per lane with in-quad transposes and 64-bit stores was measured 8 % slower: the rows are only
      // 2-byte aligned.)
      const int lane_off = (lane >> 5) * 4 * LD + (lane & 31);
      auto epi = [&](auto &lo, auto &hi) {               // arrays of the strip's tiles
        constexpr int NTS = sizeof(lo) / sizeof(lo[0]);
        phase();                                         // matrix loop | epilogue
        epi_sync();
        long bb = b0;                                    // descriptors made where they are used: see k_decrypt_m
#if NTRU_ABLATE & 512
        bb = 0;                                          // timing only: every workgroup writes the first row block (L2-resident)
#endif
        asm volatile("" : "+s"(bb));
        const long lf = (B - bb) * LD;
        const __amdgpu_buffer_rsrc_t rs_e = rows_rsrc(e + bb * LD, 2 * lf);
        const __amdgpu_buffer_rsrc_t rs_q = rows_rsrc(want_q ? quotE + bb * LD : e + bb * LD, 2 * lf);
        const unsigned char *m_l = mimg + a0m + 32 * kb0 + lane_off;
        // columns >= N (last tile only) get an offset beyond any descriptor: the hardware drops those lanes, no
        // exec-mask region per store
        int voff[NTS];
#pragma unroll
        for (int t = 0; t < NTS; t++) voff[t] = 32 * (kb0 + t) + (lane & 31) < N ? 2 * lane_off : (int)0x80000000;
        // Stores: a tile register holds row R in lanes 0-31 and row R + 4 in lanes 32-63 (32 columns each), so a store of it
        // writes two 64-byte pieces of two rows.  v_permlane32_swap exchanges the upper half of tile t's register with the
        // lower half of tile t+1's: one register then is ONE row across both tiles, 128 contiguous bytes per store (a whole
        // cache line on rows pitched to 64 elements).  What the store path pays for is the number of lines touched, not
        // the instruction count: 8-byte stores of four rows per lane group were 14 % SLOWER (profiles/r02_ablation_*).
        const int pv0 = 32 * kb0 + lane;                 // column of this lane in a tile pair starting at tile kb0
        auto out = [&](auto wq) {
#pragma unroll
          for (int j = 0; j < 4; j++) {                  // 4 rows per half-wave at a time, across the strip's tiles
            u32 mv[NTS][4];
#pragma unroll
            for (int t = 0; t < NTS; t++)
#pragma unroll
              for (int ii = 0; ii < 4; ii++) mv[t][ii] = m_l[(8 * j + ii) * LD + 32 * t];
            u32 ev[NTS][4], qv[NTS][4];
#pragma unroll
            for (int t = 0; t < NTS; t++)
#pragma unroll
              for (int ii = 0; ii < 4; ii++) {
                ev[t][ii] = (u32)(lo[t][4 * j + ii] + hi[t][4 * j + ii] + (int)mv[t][ii]) & (q - 1);
                qv[t][ii] = (u32)(0 - hi[t][4 * j + ii]) & (q - 1);
              }
#pragma unroll
            for (int t = 0; t + 1 < NTS; t += 2) {       // tile pairs: one row of 64 columns per store
              const int pvoff = pv0 + 32 * t < N ? 2 * lane : (int)0x80000000;
#pragma unroll
              for (int ii = 0; ii < 4; ii++) {
                const auto se = __builtin_amdgcn_permlane32_swap(ev[t][ii], ev[t + 1][ii], false, false);
                const int so = 2 * ((8 * j + ii) * LD + 32 * (kb0 + t));
                if (1 ABL_STORE(lo[t][4 * j])) {
                  __builtin_amdgcn_raw_buffer_store_b16((u16)se[0], rs_e, pvoff, so, ST_AUX);
                  __builtin_amdgcn_raw_buffer_store_b16((u16)se[1], rs_e, pvoff, so + 8 * LD, ST_AUX);
                  if (decltype(wq)::value) {
                    const auto sq = __builtin_amdgcn_permlane32_swap(qv[t][ii], qv[t + 1][ii], false, false);
                    __builtin_amdgcn_raw_buffer_store_b16((u16)sq[0], rs_q, pvoff, so, ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b16((u16)sq[1], rs_q, pvoff, so + 8 * LD, ST_AUX);
                  }
                }
              }
            }
            if (NTS & 1) {                               // the odd tile out: two rows of 32 columns per store
              constexpr int t = NTS - 1;
              const int so = 2 * (8 * j * LD + 32 * (kb0 + t));
              if (1 ABL_STORE(lo[t][4 * j])) {
#pragma unroll
                for (int ii = 0; ii < 4; ii++) {
                  __builtin_amdgcn_raw_buffer_store_b16((u16)ev[t][ii], rs_e, voff[t], so + 2 * ii * LD, ST_AUX);
                  if (decltype(wq)::value) __builtin_amdgcn_raw_buffer_store_b16((u16)qv[t][ii], rs_q, voff[t], so + 2 * ii * LD, ST_AUX);
                }
              }
            }
          }
          // DMA: the row loads are older than the S stores issued since and vector memory operations complete in order: at most
          // min(S, 63) outstanding = the rows have landed (and the S - 63 oldest stores with them).  At the END of the epilogue:
          // the loads have had its whole length (phase stamps: waiting after the second group of rows cost ~3 k cycles).
          constexpr int S = 16 * NTS * (decltype(wq)::value ? 2 : 1), K = S < 63 ? S : 63;
          if (DMA && dma_now && sidx == rounds - 1) __builtin_amdgcn_s_waitcnt((K & 15) | (7 << 4) | (15 << 8) | ((K >> 4) << 14));
        };
        // CHUNK: the strip's results go through the wave's LDS chunk, 8 rows at a time, laid out with the 16-byte phase the
        // rows have in global memory, and leave as ALIGNED 16-byte pieces (lane = piece: 4 rows x 16 pieces per store) plus the
        // two edges of every row segment as 2-byte stores (lane = row x element): 4 store instructions per 8 rows and array
        // instead of 4 per tile, and no 64-byte piece straddling two cache lines (bench_micro/store_pattern: 3.3 against
        // 2.4 TB/s for dense rows at N = 821).
        auto out_chunk = [&](auto wq) {
          unsigned char *oc = ochunks + wave * OC_BYTES;
          const int Wb = 2 * ((32 * NTS < N - 32 * kb0) ? 32 * NTS : N - 32 * kb0);      // bytes of a row segment of this strip
          const int LD2 = 2 * LD;
          // descriptor bases are e + bb LD (rows of this row block on): byte phase of row R's segment = (gA + 2 R LD) & 15
          const int g_e = (int)(((unsigned long long)(e + bb * LD + 32 * kb0)) & 15);
          const int g_q = (int)(((unsigned long long)((decltype(wq)::value ? quotE : e) + bb * LD + 32 * kb0)) & 15);
          const int rd = 4 * (lane >> 5), rp = lane >> 4, re = lane >> 3, sl = lane & 15, el = lane & 7;
          auto one = [&](int gA, const __amdgpu_buffer_rsrc_t &rs, auto with_m, auto val) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              // every LDS read of a step is requested before the step's LDS writes / global stores: the compiler must keep
              // the program order between the byte reads of the m image, the chunk writes and the chunk reads
              u32 mv[NTS][4];
#pragma unroll
              for (int t = 0; t < NTS; t++)
#pragma unroll
                for (int ii = 0; ii < 4; ii++) mv[t][ii] = decltype(with_m)::value ? m_l[(8 * j + ii) * LD + 32 * t] : 0u;
#pragma unroll
              for (int ii = 0; ii < 4; ii++) {             // dump rows 8 j + ii (+ 4 for the upper half-wave)
                const int a = (gA + (rd + ii) * LD2) & 15;
                unsigned char *row = oc + (rd + ii) * OC_PITCH + a + 2 * (lane & 31);
#pragma unroll
                for (int t = 0; t < NTS; t++) *(u16 *)(row + 64 * t) = (u16)val(t, 4 * j + ii, mv[t][ii]);
              }
              const int sj = 8 * j * LD2 + 64 * kb0;         // scalar part of the global byte offset: row 8 j, the strip's first column
              v4i pv[2]; int pvo[2]; u16 ev[2]; int evo[2];
#pragma unroll
              for (int half = 0; half < 2; half++) {        // aligned pieces of rows 8 j + 4 half + (lane >> 4)
                const int R = 4 * half + rp, a = (gA + R * LD2) & 15, A0 = (a + 15) & ~15, A1 = (a + Wb) & ~15;
                const int po = A0 + 16 * sl;
                pv[half] = *(const v4i *)(oc + R * OC_PITCH + po);
                pvo[half] = po + 16 <= A1 ? R * LD2 - a + po + sj : (int)0x80000000;
              }
#pragma unroll
              for (int side = 0; side < 2; side++) {        // edges of rows 8 j + (lane >> 3): head, then tail
                const int a = (gA + re * LD2) & 15, A0 = (a + 15) & ~15, A1 = (a + Wb) & ~15;
                const int eo = (side == 0 ? a : A1) + 2 * el;
                const bool ok = side == 0 ? eo < (A0 < a + Wb ? A0 : a + Wb) : (A1 >= A0 && eo < a + Wb);
                ev[side] = *(const u16 *)(oc + re * OC_PITCH + eo);
                evo[side] = ok ? re * LD2 - a + eo + sj : (int)0x80000000;
              }
              // The scalar offset is part of the vector offset: a buffer_store_dwordx4 with an SGPR soffset whose data
              // registers the NEXT instruction overwrites stores the overwritten first dword on gfx950 under load
              // (profiles/r02_hazard_store_x4_soffset.txt); the compiler only separates the two when soffset is no
              // register (tests/test_build_quality.py scans the ISA for the pattern).
              if (1 ABL_STORE(lo[0][0])) {
                __builtin_amdgcn_raw_buffer_store_b128(pv[0], rs, pvo[0], 0, ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(pv[1], rs, pvo[1], 0, ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b16(ev[0], rs, evo[0], 0, ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b16(ev[1], rs, evo[1], 0, ST_AUX);
              }
            }
          };
          one(g_e, rs_e, std::true_type{}, [&](int t, int i, u32 mm) { return (u32)(lo[t][i] + hi[t][i] + (int)mm) & (q - 1); });
          if (decltype(wq)::value) one(g_q, rs_q, std::false_type{}, [&](int t, int i, u32) { return (u32)(0 - hi[t][i]) & (q - 1); });
        };
        if (CHUNK) { if (want_q) out_chunk(std::true_type{}); else out_chunk(std::false_type{}); }
        else if (want_q) out(std::true_type{}); else out(std::false_type{});
      };
      switch (nt) {
        case 0:                                          // no strip this round: keep the barriers (and this wave's row request) in step
          phase(); epi_sync();
          if (DMA && sidx == rounds - 1) __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));   // vmcnt(0): nothing of its own is stored after them
          break;
        case 1: toeplitz_strip<M_ENC, 1>(st0, st0, tb0, tb1, g, kb0, mlow, epi, stamp_iter, 4 + 2 * sidx); break;
        case 2: toeplitz_strip<M_ENC, 2>(st0, st0, tb0, tb1, g, kb0, mlow, epi, stamp_iter, 4 + 2 * sidx); break;
        case 3: toeplitz_strip<M_ENC, 3>(st0, st0, tb0, tb1, g, kb0, mlow, epi, stamp_iter, 4 + 2 * sidx); break;
        case 4: toeplitz_strip<M_ENC, 4>(st0, st0, tb0, tb1, g, kb0, mlow, epi, stamp_iter, 4 + 2 * sidx); break;
        default: break;                                  // MAXT = 4
      }
      sidx++;
      if (sidx < rounds) phase();                        // epilogue | next matrix loop
    }, GROUPS == 2 || DMA);
  }
  if (GROUPS == 2 && group == 0) wg_barrier();        // group 1's last phase
}

__global__ __launch_bounds__(BLOCK_THREADS, 2) void k_encrypt_m(MGeom g, u32 q, const u16 *__restrict__ h,
                                                             const uint8_t *__restrict__ r,
                                                             const uint8_t *__restrict__ m, long B,
                                                             u16 *__restrict__ e, u16 *__restrict__ quotE) {
  encrypt_m_body<4, 1>(g, q, h, r, m, B, e, quotE);
}

__global__ __launch_bounds__(BLOCK_THREADS, 2) void k_encrypt_mc(MGeom g, u32 q, const u16 *__restrict__ h,
                                                              const uint8_t *__restrict__ r,
                                                              const uint8_t *__restrict__ m, long B,
                                                              u16 *__restrict__ e, u16 *__restrict__ quotE) {
  encrypt_m_body<4, 1, true>(g, q, h, r, m, B, e, quotE);
}

__global__ __launch_bounds__(BLOCK_THREADS, 2) void k_encrypt_md(MGeom g, u32 q, const u16 *__restrict__ h,
                                                              const uint8_t *__restrict__ r,
                                                              const uint8_t *__restrict__ m, long B,
                                                              u16 *__restrict__ e, u16 *__restrict__ quotE) {
  encrypt_m_body<4, 1, false, true>(g, q, h, r, m, B, e, quotE);
}

__global__ __launch_bounds__(2 * BLOCK_THREADS, 1) void k_encrypt_m8(MGeom g, u32 q, const u16 *__restrict__ h,
                                                                  const uint8_t *__restrict__ r,
                                                                  const uint8_t *__restrict__ m, long B,
                                                                  u16 *__restrict__ e, u16 *__restrict__ quotE) {
  encrypt_m_body<4, 2>(g, q, h, r, m, B, e, quotE);
}

// decryptBits on the matrix cores.  Product 1: a = f * e, e = lo7 + 128 hi (both digits non-negative, q <= 8192), planes
// [e_lo | 2 e_hi] x [f ; 64 f]; centred lift (index.js:117 verbatim); product 2: c = fp * lifted, one plane.  The lifted
// message goes from the accumulator layout (column per lane) to the operand stage (row per lane) through a 2-bit packed
// LDS image [column][8 bytes] and one expansion pass.
// GROUPS = 1: one workgroup = four waves = one row block at a time, two workgroups per CU (k_decrypt_m).
// GROUPS = 2 (k_decrypt_m8): ONE workgroup of eight waves per CU = two groups of four, each with its own row blocks, stages and
// packed image, sharing the key arrays and the lift table.  Every matrix loop and every epilogue is a PHASE between two
// workgroup barriers, and group 1 runs one phase behind group 0: while one group's waves are in their matrix loops, the
// other group's waves (their partners on the SIMDs) are in an epilogue / staging phase, by construction instead of by
// luck.  Phases per row block: stage, then (loop, epilogue) per strip and product, with the image expansion between the
// products: 10 at N = 821 -- an even number, so the two groups stay in opposite phases.
// DMA (k_decrypt_m8d, GROUPS = 2 only): the e rows of the NEXT row block are requested with direct global -> LDS loads
// (buffer_load_dwordx4 ... lds: no registers) at the start of the last epilogue of the current one, i.e. IN FRONT of that
// epilogue's store burst instead of behind it (phase stamps: rows requested behind the stores come back 8-9 k cycles later).
// Layout: every row has ONE slot of rp = 64 NT + 16 bytes that first receives the raw row (2N bytes, whatever its alignment in
// global memory: the loads place lane l's 16 bytes at slot + 16 l) and then, converted in place by the wave that owns the
// row, its two digit planes (low at +0, high at +pitchA): no other wave touches the slot, no extra barrier.  The mod-p tables
// of product 2 cannot live on the e_hi stage any more (it is being filled during product 2's last epilogue): they are built
// once, at LDS address 0, for both groups.  The loads are waited for with s_waitcnt vmcnt(K), K = the number of stores issued
// behind them so far (vector memory operations complete in order), after the first rows of that epilogue have been stored.
static __host__ __device__ inline int dec_dma_row_pitch(int NT) { return 64 * NT + 16; }          // 16 (4 NT + 1): an odd multiple of 16
static __host__ __device__ inline int dec_dma_m3_bytes(int N, int p) {
  const int span = (p - 1) * (p - 1) * N;
  return (((span + 4) & ~3) + span + 1 + 15) & ~15;
}

template <int GROUPS, bool DMA = false>
static __device__ __forceinline__ void decrypt_m_body(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                      const uint8_t *__restrict__ fp,
                                                      const u16 *__restrict__ e, long B,
                                                      uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                      u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // DMA: a barrier that orders LDS traffic only.  With direct-to-LDS loads in flight the compiler puts s_waitcnt vmcnt(0) in front of
  // every __syncthreads() -- which also waits for every outstanding STORE, the very queue the early loads are meant to get ahead of.
  // The loads are waited for explicitly (vmcnt(K), see the last epilogue), everything else these barriers order lives in LDS.
  auto wg_barrier = [&]() {
    if (DMA) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  };
  const int group = GROUPS == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  // LDS layout: per group [e_hi stage][e_lo stage][packed image], then the shared key arrays and the lift table.  Group 0's
  // e_hi stage is at LDS address 0: the mod-p tables of product 2 are overlaid on it and their lookups need no base add.
  static_assert(!DMA || GROUPS == 2, "the direct-to-LDS variant is the lock-step kernel");
  const int RP = DMA ? dec_dma_row_pitch(g.NT) : g.pitchA;                       // row pitch of the operand stage(s)
  const int m3b = DMA ? dec_dma_m3_bytes(g.N, (int)p) : 0;
  const int gbytes = (DMA ? 32 * RP : 64 * g.pitchA) + 256 * g.NT;
  unsigned char *stHi = DMA ? lds + m3b + group * gbytes + g.pitchA : lds + group * gbytes;   // DMA: the high plane of row R at slot R + pitchA
  unsigned char *stLo = DMA ? lds + m3b + group * gbytes : stHi + 32 * g.pitchA;
  unsigned char *blp = DMA ? stLo + 32 * RP : stLo + 32 * g.pitchA;              // [8 row groups][32 NT columns]: 4 rows x 2 bits per byte
  u32 *TF = (u32 *)(lds + m3b + GROUPS * gbytes), *TP = TF + 4 * g.tpitch;
  unsigned char *lift_lut = (unsigned char *)(TP + 4 * g.tpitch);   // [q]: centred lift followed by mod p, index.js:117 verbatim
  // mod-p tables of product 2, rebuilt per row block once the e stages are dead: (-x) mod p at LDS address x, so that
  // the quotient lookup's address IS the `high` accumulator; x mod p at M3V + x, the base folded into the low + high add
  const int M3V = __builtin_amdgcn_readfirstlane(((int)((p - 1) * (p - 1)) * g.N + 4) & ~3);   // both tables inside the e_hi stage
  unsigned char *m3_lut = DMA ? lds : stHi;
  const int tid0 = threadIdx.x & (BLOCK_THREADS - 1), lane0 = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  if (DMA)
    for (int x = (int)threadIdx.x; x <= (int)((p - 1) * (p - 1)) * g.N; x += GROUPS * BLOCK_THREADS) {
      const u32 rm = mod_small((u32)x, p);
      m3_lut[x] = (unsigned char)(rm ? p - rm : 0u);
      m3_lut[M3V + x] = (unsigned char)rm;
    }
  build_toeplitz_array(TF, g, [&](int i) { return (int)f[i]; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
  build_toeplitz_array(TP, g, [&](int i) { return (int)fp[i]; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
  for (u32 x = threadIdx.x; x < q; x += GROUPS * BLOCK_THREADS) lift_lut[x] = (unsigned char)mod_small(2 * x > q ? x + 1 : x, p);
#ifndef NTRU_PHASE_MASK
#define NTRU_PHASE_MASK 15       // which boundaries of the lock-step schedule are barriers (tuning experiments): 1 = product 1 loop | epilogue,
#endif                           // 2 = product 1 epilogue | next loop, 4 / 8 = the same for product 2
  auto phase = [&](int kind) { if (GROUPS == 2 && (NTRU_PHASE_MASK & kind)) wg_barrier(); };
  if (GROUPS == 2 && group == 1) wg_barrier();                      // group 1 runs one phase behind group 0
  const bool want_q1 = quot1 != nullptr, want_r1 = rem1 != nullptr, want_q2 = quot2 != nullptr;
  const long nrb = (B + 31) >> 5;
  const int nch = 2 * g.NT;
  const u32 qm2 = (q - 1) * 0x00010001u;
  int sidx = 0, stamp_iter = -1;
  const long stride = (long)gridDim.x * GROUPS, iters = (nrb + stride - 1) / stride;
  const int rounds = (((g.NT + 3) >> 2) + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  // DMA: rows wave, wave + 4, ... of row block rbx into their slots, 2 N bytes each as 16-byte pieces (the last piece reads
  // up to 14 bytes of the next row or, at the end of the batch, zeros: those columns are masked when the row is converted)
  auto dma_rows = [&](long rbx, int lane) {
    // Pieces start at ABSOLUTELY dword-aligned addresses (descriptor based at the dword at or below the row block, size in whole
    // dwords: the range check is per dword, and a dword that holds the last coefficient of the batch must not count as out of
    // range; the two bytes it may read past the end lie in the same aligned dword as that coefficient).  A row therefore
    // lands in its slot 0 or 2 bytes in; the conversion shifts by that (wave-uniform) amount.
    const long b0x = rbx << 5 < B ? rbx << 5 : B;
    const unsigned long long a = (unsigned long long)(e + b0x * g.ld);
    const int a0 = (int)(a & 3);
    const __amdgpu_buffer_rsrc_t rs = rows_rsrc((const void *)(a & ~3ULL), (2 * (B - b0x) * g.ld + a0 + 3) & ~3L);
#pragma unroll
    for (int j = 0; j < 32 / WAVES_PER_BLOCK; j++) {
      const int row = wave + WAVES_PER_BLOCK * j, ro = a0 + row * 2 * g.ld;
      const int npc = ((ro & 3) + 2 * g.N + 15) >> 4;                             // 16-byte pieces of this row
#pragma unroll
      for (int half = 0; half < 2; half++)
        if (lane + 64 * half < npc)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(stLo + row * RP + 1024 * half), 16,
                                                   (ro & ~3) + 1024 * half + 16 * lane, 0, 0, 0);
    }
  };
  if (DMA) {
    long rb0 = (long)blockIdx.x * GROUPS + group;
    dma_rows(rb0 < nrb ? rb0 : nrb, lane0);
    __builtin_amdgcn_s_waitcnt(0);                                                // nothing is in flight besides them yet
  }
  for (long it = 0; it < iters; it++) {
    // a group without a row block left (the last trip of an odd count) still walks through every phase: its row block is
    // placed at the end of the batch, where every load reads zero and every store is dropped by the buffer descriptors
    long rb = (long)blockIdx.x * GROUPS + group + it * stride;
    rb = rb < nrb ? rb : nrb;
    long rb_next = (long)blockIdx.x * GROUPS + group + (it + 1) * stride;
    rb_next = rb_next < nrb ? rb_next : nrb;
    stamp_iter++;
    STAMP(0);
    int lane = lane0, N = g.N, LD = g.ld;                // see k_encrypt_m
    asm volatile("" : "+v"(lane), "+s"(N), "+s"(LD));
    const u32 *tbf = frag_lane_base(TF, g, lane), *tbp = frag_lane_base(TP, g, lane);
    const unsigned char *st0 = stLo + (lane & 31) * RP + 16 * (lane >> 5);
    const unsigned char *st1 = stHi + (lane & 31) * RP + 16 * (lane >> 5);
    u32 mlow[4];
    diag_low_mask(lane, mlow);
    const long b0 = rb << 5 < B ? rb << 5 : B, left = (B - b0) * LD;
    const AlignedSrc src_e = aligned_src(e + b0 * LD, 2 * left);
    // lane = 16 coefficients.  (Requesting these loads before the barrier, as k_encrypt_m does, measured 3 % slower with two
    // workgroups per CU and 5 % slower in the lock-step schedule.)
    constexpr int RPW = 32 / WAVES_PER_BLOCK;
    RawChunks<2> raw[RPW];
    int sh[RPW];
    auto request_rows = [&]() {
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int pos0 = src_e.a0 + 2 * (wave + WAVES_PER_BLOCK * j) * LD;
        sh[j] = __builtin_amdgcn_readfirstlane(pos0 & 15);
#if defined(NTRU_ABLATE) && (NTRU_ABLATE & 4)
        raw[j] = fake_raw<2>(pos0 + lane);
#else
        raw[j] = load_raw<2>(src_e, pos0 + 32 * lane, sh[j]);
#endif
      }
    };
    wg_barrier();
    STAMP(1);
    if (!DMA) request_rows();
    if (DMA) {                                           // the rows are in their slots (every wave waited for its own loads)
      const int a0 = (int)((unsigned long long)(e + b0 * LD) & 3);
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int row = wave + WAVES_PER_BLOCK * j;
        const unsigned char *slot = stLo + row * RP + 32 * (lane < 2 * g.NT ? lane : 0);
        raw[j].c[0] = *(const v4i *)slot;
        raw[j].c[1] = *(const v4i *)(slot + 16);
        raw[j].tail = *(const u32 *)(slot + 32);
        sh[j] = __builtin_amdgcn_readfirstlane((a0 + row * 2 * LD) & 3);
      }
    }
    {
      const int c16 = lane;
      u32 cmask[8];                                      // columns >= N of the last chunk(s) are zero; coefficients mod q
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int left2 = N - (16 * c16 + 2 * c);
        cmask[c] = qm2 & (left2 >= 2 ? 0xFFFFFFFFu : (left2 == 1 ? 0x0000FFFFu : 0u));
      }
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int row = wave + WAVES_PER_BLOCK * j;
        v4i v[2];                                        // 16 coefficients as u16 pairs
        shift_raw<2>(raw[j], sh[j], v);
        u32 x[8];
#pragma unroll
        for (int c = 0; c < 4; c++) { x[c] = (u32)v[0][c]; x[4 + c] = (u32)v[1][c]; }
        u32 lo[4], hi[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const u32 xa = x[2 * c] & cmask[2 * c], xb = x[2 * c + 1] & cmask[2 * c + 1];
          lo[c] = __builtin_amdgcn_perm(xb & 0x007F007Fu, xa & 0x007F007Fu, 0x06040200u);
          hi[c] = __builtin_amdgcn_perm((xb >> 6) & 0x00FE00FEu, (xa >> 6) & 0x00FE00FEu, 0x06040200u);
        }
        if (c16 < nch) {
          *(uint4 *)(stLo + row * RP + 16 * c16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
          *(uint4 *)(stHi + row * RP + 16 * c16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        }
      }
    }
    STAMP(2);
    wg_barrier();
    STAMP(3);
    const int lane_off = (lane >> 5) * 4 * LD + (lane & 31);
    // ---- product 1: a = f * e mod q; witness stores; lifted message -> packed image
    sidx = 0;
    for_each_strip<4>(g.NT, GROUPS == 2 ? wave ^ (2 * group) ^ (2 * blockIdx.x >= gridDim.x ? 2 : 0) : wave, [&](int kb0, int nt) {
      auto epi = [&](auto &lo, auto &hi) {
        constexpr int NTS = sizeof(lo) / sizeof(lo[0]);
        phase(1);                                        // matrix loop | epilogue
        // descriptors are made here, from a re-materialised row-block base, so that they live in scalar registers only
        // while they are used (held across the matrix loops they are spilled to VGPRs and every store becomes a
        // waterfall loop)
        long bb = b0;
        asm volatile("" : "+s"(bb));
        const long lf = (B - bb) * LD;
        const __amdgpu_buffer_rsrc_t rs_r1 = rows_rsrc(want_r1 ? rem1 + bb * LD : nullptr, want_r1 ? 2 * lf : 0);
        const __amdgpu_buffer_rsrc_t rs_q1 = rows_rsrc(want_q1 ? quot1 + bb * LD : nullptr, want_q1 ? 2 * lf : 0);
        int voff[NTS];                                   // see k_encrypt_m
#pragma unroll
        for (int t = 0; t < NTS; t++) voff[t] = 32 * (kb0 + t) + (lane & 31) < N ? 2 * lane_off : (int)0x80000000;
        auto out = [&](auto wr, auto wq) {
#pragma unroll
          for (int j = 0; j < 4; j++) {                  // 4 rows x the strip's tiles at a time: remainders (and their
            u32 xs[NTS][4], lv[NTS][4];                  // stores), then their table lookups in flight together, then packing
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
              const int i = 4 * j + ii, ro = ii + 8 * j;
#pragma unroll
              for (int t = 0; t < NTS; t++) {
                const u32 x = (u32)(lo[t][i] + hi[t][i]) & (q - 1);
                xs[t][ii] = x;
                const int so = 2 * (ro * LD + 32 * (kb0 + t));
                if (1 ABL_STORE(lo[t][i])) {
#if NTRU_ABLATE & 32768
                  __builtin_amdgcn_raw_buffer_store_b16((u16)(x ^ (u32)hi[t][i]), rs_r1, voff[t], so, ST_AUX);       // timing only: one store
#else
                  if (decltype(wr)::value) __builtin_amdgcn_raw_buffer_store_b16((u16)x, rs_r1, voff[t], so, ST_AUX);
                  if (decltype(wq)::value) __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi[t][i]) & (q - 1)), rs_q1, voff[t], so, ST_AUX);
#endif
                }
              }
            }
#pragma unroll
            for (int t = 0; t < NTS; t++)
#pragma unroll
#if NTRU_ABLATE & 16384
              for (int ii = 0; ii < 4; ii++) lv[t][ii] = xs[t][ii] & 1u;                       // timing only: no lift lookups
#else
              for (int ii = 0; ii < 4; ii++) lv[t][ii] = lift_lut[xs[t][ii]];
#endif
#pragma unroll
            for (int t = 0; t < NTS; t++) {
              const int col = 32 * (kb0 + t) + (lane & 31);
              const u32 pk = lv[t][0] | (lv[t][1] << 2) | (lv[t][2] << 4) | (lv[t][3] << 6);
              blp[((lane >> 5) + 2 * j) * 32 * g.NT + col] = (unsigned char)(col < N ? pk : 0u);   // [row group 2j+hh][column]
            }
          }
        };
        if (want_r1 && want_q1) out(std::true_type{}, std::true_type{});
        else if (want_r1) out(std::true_type{}, std::false_type{});
        else if (want_q1) out(std::false_type{}, std::true_type{});
        else out(std::false_type{}, std::false_type{});
      };
      switch (nt) {
        case 0: phase(1); break;                         // (the strip list hands out empty strips only to keep the phases in step)
        case 1: toeplitz_strip<M_DEC1, 1>(st0, st1, tbf, tbf, g, kb0, mlow, epi, stamp_iter, 4 + 2 * sidx); break;
        case 2: toeplitz_strip<M_DEC1, 2>(st0, st1, tbf, tbf, g, kb0, mlow, epi, stamp_iter, 4 + 2 * sidx); break;
        case 3: toeplitz_strip<M_DEC1, 3>(st0, st1, tbf, tbf, g, kb0, mlow, epi, stamp_iter, 4 + 2 * sidx); break;
        default: toeplitz_strip<M_DEC1, 4>(st0, st1, tbf, tbf, g, kb0, mlow, epi, stamp_iter, 4 + 2 * sidx); break;
      }
      sidx++;
      if (sidx < rounds) phase(2);                       // epilogue | next matrix loop (after the last strip: the barrier below)
    }, GROUPS == 2);
    wg_barrier();                                    // every wave is done with the e stages; packed image complete
    STAMP(8);
#if !(NTRU_ABLATE & 64)
    if (!DMA) for (int x = tid0; x <= (int)((p - 1) * (p - 1)) * N; x += BLOCK_THREADS) {
      const u32 rm = mod_small((u32)x, p);
      m3_lut[x] = (unsigned char)(rm ? p - rm : 0u);
      m3_lut[M3V + x] = (unsigned char)rm;
    }
#endif
#if !(NTRU_ABLATE & 32)
    {   // packed image -> byte stage: all of a wave's reads in flight before the first write (as a read-write loop this
        // pass was one LDS round trip per dword: 7 k cycles per row block in the phase stamps)
      constexpr int RPW = 32 / WAVES_PER_BLOCK;
      u32 pv[RPW][4];
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int row = wave + WAVES_PER_BLOCK * j, rgb = 2 * (row >> 3) + ((row >> 2) & 1);
        const u32 *src = (const u32 *)(blp + rgb * 32 * g.NT);
#pragma unroll
        for (int it = 0; it < 4; it++) pv[j][it] = src[(lane + 64 * it) < 8 * g.NT ? lane + 64 * it : 0];
      }
#pragma unroll
      for (int j = 0; j < RPW; j++) {
        const int row = wave + WAVES_PER_BLOCK * j, sh = 2 * (row & 3);
#pragma unroll
        for (int it = 0; it < 4; it++)
          if (lane + 64 * it < 8 * g.NT) *(u32 *)(stLo + row * RP + 4 * (lane + 64 * it)) = (pv[j][it] >> sh) & 0x03030303u;
      }
    }
#endif
    STAMP(9);
    wg_barrier();
    STAMP(10);
    // ---- product 2: c = fp * lifted mod p
    sidx = 0;
    for_each_strip<4>(g.NT, GROUPS == 2 ? wave ^ (2 * group) ^ (2 * blockIdx.x >= gridDim.x ? 2 : 0) : wave, [&](int kb0, int nt) {
      auto epi = [&](auto &lo, auto &hi) {
        constexpr int NTS = sizeof(lo) / sizeof(lo[0]);
        phase(4);                                        // matrix loop | epilogue
        // DMA: this barrier is behind every wave's last loop of the row block when sidx is the last round: the stage slots are
        // free, and nothing of this epilogue has been stored yet
        const bool dma_now = DMA && sidx == rounds - 1 && it + 1 < iters;
        if (dma_now) dma_rows(rb_next, lane);
        long bb = b0;                                    // see product 1
        asm volatile("" : "+s"(bb));
        const long lf = (B - bb) * LD;
        const __amdgpu_buffer_rsrc_t rs_v = rows_rsrc(value + bb * LD, lf);
        const __amdgpu_buffer_rsrc_t rs_q2 = rows_rsrc(want_q2 ? quot2 + bb * LD : nullptr, want_q2 ? lf : 0);
        int voff[NTS];                                   // see k_encrypt_m
#pragma unroll
        for (int t = 0; t < NTS; t++) voff[t] = 32 * (kb0 + t) + (lane & 31) < N ? lane_off : (int)0x80000000;
        auto out = [&](auto wq) {
#pragma unroll
          for (int j = 0; j < 4; j++) {                  // lookups of 4 rows x the strip's tiles in flight before their stores
            u32 va[NTS][4], vb[NTS][4];
#pragma unroll
            for (int t = 0; t < NTS; t++)
#pragma unroll
              for (int ii = 0; ii < 4; ii++) {
#if NTRU_ABLATE & 4096
                va[t][ii] = (u32)(lo[t][4 * j + ii] + hi[t][4 * j + ii]); vb[t][ii] = (u32)hi[t][4 * j + ii];   // timing only: no lookups
#else
                va[t][ii] = m3_lut[(u32)(lo[t][4 * j + ii] + hi[t][4 * j + ii] + M3V)];
                vb[t][ii] = decltype(wq)::value ? (u32)m3_lut[(u32)hi[t][4 * j + ii]] : 0u;
#endif
              }
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
#pragma unroll
              for (int t = 0; t < NTS; t++) {
                const int so = (ii + 8 * j) * LD + 32 * (kb0 + t);
                if (1 ABL_STORE(lo[t][4 * j + ii])) {
#if NTRU_ABLATE & 8192
                  __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(va[t][ii] ^ vb[t][ii]), rs_v, voff[t], so, ST_AUX);   // timing only: one store
#else
                  __builtin_amdgcn_raw_buffer_store_b8((uint8_t)va[t][ii], rs_v, voff[t], so, ST_AUX);
                  if (decltype(wq)::value) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)vb[t][ii], rs_q2, voff[t], so, ST_AUX);
#endif
                }
              }
            }
          }
          // DMA: the row loads are older than the S stores issued since (vector memory operations complete in order): at most
          // min(S, 63) outstanding operations = the loads have landed.  At the end of the epilogue: they have had its whole length.
          constexpr int S = 16 * NTS * (decltype(wq)::value ? 2 : 1), K = S < 63 ? S : 63;
          if (DMA && dma_now) __builtin_amdgcn_s_waitcnt((K & 15) | (7 << 4) | (15 << 8) | ((K >> 4) << 14));
        };
        if (want_q2) out(std::true_type{}); else out(std::false_type{});
      };
      switch (nt) {
        case 0:                                          // no strip this round: the phase barrier -- and this wave's rows of the next row block
          phase(4);
          if (DMA && sidx == rounds - 1 && it + 1 < iters) {
            dma_rows(rb_next, lane);
            __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));                       // vmcnt(0): it stores nothing behind them
          }
          break;
        case 1: toeplitz_strip<M_DEC2, 1>(st0, st0, tbp, tbp, g, kb0, mlow, epi, stamp_iter, 11 + 2 * sidx); break;
        case 2: toeplitz_strip<M_DEC2, 2>(st0, st0, tbp, tbp, g, kb0, mlow, epi, stamp_iter, 11 + 2 * sidx); break;
        case 3: toeplitz_strip<M_DEC2, 3>(st0, st0, tbp, tbp, g, kb0, mlow, epi, stamp_iter, 11 + 2 * sidx); break;
        default: toeplitz_strip<M_DEC2, 4>(st0, st0, tbp, tbp, g, kb0, mlow, epi, stamp_iter, 11 + 2 * sidx); break;
      }
      sidx++;
      if (sidx < rounds) phase(8);
    }, GROUPS == 2);
  }
  if (GROUPS == 2 && group == 0) wg_barrier();        // group 1's last phase
}

__global__ __launch_bounds__(BLOCK_THREADS, 2) void k_decrypt_m(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                             const uint8_t *__restrict__ fp,
                                                             const u16 *__restrict__ e, long B,
                                                             uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                             u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  decrypt_m_body<1>(g, q, p, f, fp, e, B, value, quot1, rem1, quot2);
}

__global__ __launch_bounds__(2 * BLOCK_THREADS, 1) void k_decrypt_m8(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                                  const uint8_t *__restrict__ fp,
                                                                  const u16 *__restrict__ e, long B,
                                                                  uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                                  u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  decrypt_m_body<2>(g, q, p, f, fp, e, B, value, quot1, rem1, quot2);
}

__global__ __launch_bounds__(2 * BLOCK_THREADS, 1) void k_decrypt_m8d(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                                   const uint8_t *__restrict__ fp,
                                                                   const u16 *__restrict__ e, long B,
                                                                   uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                                   u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  decrypt_m_body<2, true>(g, q, p, f, fp, e, B, value, quot1, rem1, quot2);
}

// ---- family 4, role-split variants ----------------------------------------------------------------------------------
// k_encrypt_m / k_decrypt_m above give every wave the whole job of its column strips: stage, matrix loops, epilogue
// arithmetic, table lookups and 2-byte result stores, phase after phase; the two co-resident workgroups of a CU overlap
// almost none of it (the ablation times are additive).  The role-split kernels run ONE workgroup
// of eight waves per CU, two per SIMD with complementary jobs at all times:
//   waves 0-3, MATRIX waves: the strip loops of toeplitz_strip and nothing else -- their "epilogue" is one add and one
//     ds_write_b32 per accumulator register: the raw pair (low + high | high << 16) goes into an LDS chunk;
//   waves 4-7, IO waves: everything that touches global memory.  They stage the next row block's operand while the
//     matrix waves compute, and they DRAIN the chunks: reduce modulo q / add m / negate with packed 16-bit arithmetic
//     on 8 coefficients per lane and store them as 16-byte pieces that are ALIGNED IN GLOBAL MEMORY.  Rows of N odd
//     elements start at every alignment, so a chunk row is laid out with the same misalignment as its row of the
//     result array (a_row = byte address of the row start mod 16, which does not depend on the row block): aligned LDS
//     reads then are aligned global pieces; the <= 7 elements on either side of a row segment go out one by one.
// Per round (four adjacent strips): matrix loops || drain of the previous round's chunks, barrier, dump || staging,
// barrier.  Preconditions checked by the host: every batch array 16-byte aligned, LDS fits; otherwise the kernels above.
static __host__ __device__ inline int m2_rounds(int NT) { return (((NT + 3) >> 2) + 3) >> 2; }
// Strip j of the 4 * rounds strips (sizes as even as possible, in column order): first tile and number of tiles.
static __host__ __device__ inline void m2_strip(int NT, int j, int *kb0, int *nt) {
  const int n_str = 4 * m2_rounds(NT), base = NT / n_str, rem = NT % n_str;
  *nt = base + (j < rem ? 1 : 0);
  *kb0 = j * base + (j < rem ? j : rem);
}
// Chunk row pitch of matrix wave w: 4 bytes per coefficient of its widest strip + room for twice the misalignment.
static __host__ __device__ inline int m2_chunk_pitch(int NT, int w) {
  int kb0, nt;
  m2_strip(NT, w, &kb0, &nt);                            // the strips of round 0 are the widest ones of each wave
  return 128 * nt + 32;
}

// Matrix wave: raw results of a strip into its chunk.  Register i of a tile is row (i & 3) + 8 (i >> 2) + 4 (lane >> 5),
// column lane & 31 (see k_encrypt_m); chunk element (row, c) lives at row * cp + 2 a_row + 4 c.
template <class Acc>
static __device__ __forceinline__ void m2_dump(unsigned char *chunk, int cp, int LD, int lane, const Acc &lo, const Acc &hi) {
  constexpr int NTS = sizeof(lo) / sizeof(lo[0]);
  u32 addr[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    addr[i] = (u32)(row * cp + 2 * ((2 * row * LD) & 15) + 4 * (lane & 31));
  }
#pragma unroll
  for (int t = 0; t < NTS; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 pr = __builtin_amdgcn_perm((u32)hi[t][i], (u32)(lo[t][i] + hi[t][i]), 0x05040100u);   // (low + high) & 0xffff | high << 16
      *(u32 *)(chunk + addr[i] + 128 * t) = pr;
    }
}

// Geometry of a chunk drain (shared by every role-split kernel).  A row segment of `ncol` coefficients occupies the bytes
// [a, a + L) of the row's "aligned space" (byte 0 = the 16-byte boundary at or below the segment's first element in a
// uint16 result array; the same boundary is an 8-byte boundary of a byte array), L = 2 ncol.  Full 16-byte pieces k are
// handled 16 lanes per row, 4 rows per pass, 8 passes; the elements before the first / after the last full piece 16
// lanes per row as well (8 head slots, 8 tail slots).
struct M2Piece { bool ok; int row, k, a; long at; };      // at: element index of the piece's first coefficient
struct M2Edge { bool ok; int row, y, a; long at; };       // y: byte position in the aligned space
static __device__ __forceinline__ M2Piece m2_piece(int it, int lane, int L, int LD, int kb0, long b0, long B) {
  M2Piece p;
  p.row = 4 * it + (lane >> 4);
  p.a = (2 * p.row * LD) & 15;
  const int kf0 = (p.a + 15) >> 4, kf1 = (p.a + L) >> 4;
  p.k = kf0 + (lane & 15);
  p.ok = b0 + p.row < B && p.k < kf1;
  p.at = (b0 + p.row) * LD + 32 * kb0 + ((16 * p.k - p.a) >> 1);
  return p;
}
static __device__ __forceinline__ M2Edge m2_edge(int it, int lane, int L, int LD, int kb0, long b0, long B) {
  M2Edge g;
  g.row = 4 * it + (lane >> 4);
  g.a = (2 * g.row * LD) & 15;
  const int j = lane & 15, kf0 = (g.a + 15) >> 4, kf1 = (g.a + L) >> 4;
  const int head_end = 16 * kf0 < g.a + L ? 16 * kf0 : g.a + L;             // head = [a, head_end)
  const int tail_beg = 16 * kf1 > head_end ? 16 * kf1 : head_end;           // tail = [tail_beg, a + L)
  g.y = j < 8 ? g.a + 2 * j : tail_beg + 2 * (j - 8);
  g.ok = b0 + g.row < B && (j < 8 ? g.y < head_end : g.y < g.a + L);
  g.at = (b0 + g.row) * LD + 32 * kb0 + ((g.y - g.a) >> 1);
  return g;
}
static __device__ __forceinline__ int m2_seg_bytes(int kb0, int nt, int N) {
  const int ncol = 32 * nt < N - 32 * kb0 ? 32 * nt : N - 32 * kb0;
  return 2 * (ncol > 0 ? ncol : 0);
}

// IO wave, encrypt: the plaintext bytes a chunk drain will add, requested a whole phase ahead (every load of the drain in
// flight at once: as a load per pass the drain was one HBM round trip per pass, 3 ms per 2^20 instead of 1.5).
struct M2EncPre { uint2 m8[8]; u32 mb[8]; };
static __device__ __forceinline__ void m2_prefetch_encrypt(M2EncPre &pre, int kb0, int nt, long b0, long B, int N, int LD,
                                                           const uint8_t *__restrict__ m, int lane) {
  const int L = m2_seg_bytes(kb0, nt, N);
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const M2Piece p = m2_piece(it, lane, L, LD, kb0, b0, B);
    pre.m8[it] = p.ok ? *(const uint2 *)(m + p.at) : make_uint2(0u, 0u);   // 8-byte aligned: 2 at is a multiple of 16
    const M2Edge g = m2_edge(it, lane, L, LD, kb0, b0, B);
    pre.mb[it] = g.ok ? (u32)m[g.at] : 0u;
  }
}

// IO wave: one chunk of an encrypt strip -> e = (raw + m) mod q and quotientE = -high mod q.
template <bool WQ>
static __device__ __forceinline__ void m2_drain_encrypt(const unsigned char *chunk, int cp, int kb0, int nt, long b0, long B,
                                                        int N, int LD, u32 q, const M2EncPre &pre,
                                                        u16 *__restrict__ e, u16 *__restrict__ quotE, int lane) {
  const int L = m2_seg_bytes(kb0, nt, N);
  if (L <= 0) return;
  const u32 qm2 = (q - 1) * 0x00010001u;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const M2Piece p = m2_piece(it, lane, L, LD, kb0, b0, B);
    if (p.ok) {
      const unsigned char *src = chunk + p.row * cp + 32 * p.k;
      const v4i x0 = *(const v4i *)src, x1 = *(const v4i *)(src + 16);
      const uint2 mv = pre.m8[it];
      u32 rm[4], hv[4];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        rm[c] = __builtin_amdgcn_perm((u32)x0[2 * c + 1], (u32)x0[2 * c], 0x05040100u);
        hv[c] = __builtin_amdgcn_perm((u32)x0[2 * c + 1], (u32)x0[2 * c], 0x07060302u);
        rm[2 + c] = __builtin_amdgcn_perm((u32)x1[2 * c + 1], (u32)x1[2 * c], 0x05040100u);
        hv[2 + c] = __builtin_amdgcn_perm((u32)x1[2 * c + 1], (u32)x1[2 * c], 0x07060302u);
      }
      const u32 mm[4] = {__builtin_amdgcn_perm(0u, mv.x, 0x0c010c00u), __builtin_amdgcn_perm(0u, mv.x, 0x0c030c02u),
                         __builtin_amdgcn_perm(0u, mv.y, 0x0c010c00u), __builtin_amdgcn_perm(0u, mv.y, 0x0c030c02u)};
      uint4 ev, qv;
      u32 *evp = (u32 *)&ev, *qvp = (u32 *)&qv;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        evp[c] = as_u32(as_pair(rm[c]) + as_pair(mm[c])) & qm2;
        qvp[c] = as_u32((u16x2){0, 0} - as_pair(hv[c])) & qm2;
      }
      *(uint4 *)(e + p.at) = ev;
      if (WQ) *(uint4 *)(quotE + p.at) = qv;
    }
  }
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const M2Edge g = m2_edge(it, lane, L, LD, kb0, b0, B);
    if (g.ok) {
      const u32 pr = *(const u32 *)(chunk + g.row * cp + 2 * g.y);
      e[g.at] = (u16)((pr + pre.mb[it]) & (q - 1));
      if (WQ) quotE[g.at] = (u16)((0u - (pr >> 16)) & (q - 1));
    }
  }
}

// encryptBits, role-split (see above).  Grid = one workgroup per CU; LDS: key arrays, TWO r stages, four chunks.
__global__ __launch_bounds__(512, 1) void k_encrypt_m2(MGeom g, u32 q, const u16 *__restrict__ h,
                                                         const uint8_t *__restrict__ r, const uint8_t *__restrict__ m,
                                                         long B, u16 *__restrict__ e, u16 *__restrict__ quotE) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u32 *T0 = (u32 *)lds, *T1 = T0 + 4 * g.tpitch;
  unsigned char *stR = (unsigned char *)(T1 + 4 * g.tpitch);       // two stages of 32 rows
  unsigned char *chunks = stR + 2 * 32 * g.pitchA;
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave8 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const bool io = wave8 >= 4;
  const int w4 = wave8 & 3;
  int chunk_off = 0;
  for (int w = 0; w < w4; w++) chunk_off += 32 * m2_chunk_pitch(g.NT, w);
  unsigned char *chunk = chunks + chunk_off;                         // written by matrix wave w4, drained by io wave w4
  const int cp = m2_chunk_pitch(g.NT, w4);
  const int hthr = (int)(q >> 1) - 65;
  auto hs_of = [&](int i) { int hv = (int)(h[i] & (q - 1)); return hv > hthr ? hv - (int)q : hv; };
  build_toeplitz_array(T0, g, [&](int i) { const int hs = hs_of(i); return ((hs + 64) & 127) - 64; }, tid0, 512);
  build_toeplitz_array(T1, g, [&](int i) { const int hs = hs_of(i); const int d0 = ((hs + 64) & 127) - 64; return ((hs - d0) >> 7) * 4; }, tid0, 512);
  const bool want_q = quotE != nullptr;
  const long nrb = (B + 31) >> 5;
  const int rounds = m2_rounds(g.NT);
  const int stamp_iter = 0;
  (void)stamp_iter;

  // io: rows w4, w4 + 4, ... of row block rb into stage `buf` (r in {0..3} bytes; columns >= N zero), in two halves so
  // that the loads are in flight while the wave drains a chunk
  constexpr int RPW = 8;
  RawChunks<1> in_r[RPW];
  auto stage_load = [&](long rb) {
    int lane = lane0, LD = g.ld;
    asm volatile("" : "+v"(lane), "+s"(LD));
    const long b0 = rb << 5, left = (B - b0) * LD;
    const AlignedSrc src_r = aligned_src(r + b0 * LD, left);
#pragma unroll
    for (int j = 0; j < RPW; j++) in_r[j] = load_raw<1>(src_r, src_r.a0 + (w4 + 4 * j) * LD + 16 * lane, 0);
  };
  auto stage_store = [&](long rb, int buf) {
    int lane = lane0, N = g.N, LD = g.ld;
    asm volatile("" : "+v"(lane), "+s"(N), "+s"(LD));
    const int a0 = (int)((unsigned long long)(r + (rb << 5) * LD) & 15);
    const v4i mk = col_mask16(16 * lane, N);
    unsigned char *st = stR + buf * 32 * g.pitchA;
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int row = w4 + 4 * j;
      v4i v[1];
      shift_raw<1>(in_r[j], a0 + row * LD, v);
      if (lane < 2 * g.NT) *(v4i *)(st + row * g.pitchA + 16 * lane) = v[0] & mk;
    }
  };

  if (io && (long)blockIdx.x < nrb) { stage_load(blockIdx.x); stage_store(blockIdx.x, 0); }
  __syncthreads();                                                   // key arrays and the first stage are in place
  // The two roles run separate copies of the same (row block, round) loop -- same barrier sequence, but what one role
  // keeps across iterations (prefetched operand rows, plaintext bytes) is not live in the other role's code.
  if (!io) {
    int it = 0;
    for (long rb = blockIdx.x; rb < nrb; rb += gridDim.x, it++) {
      const int buf = it & 1;
      for (int rho = 0; rho < rounds; rho++) {
        int kb0, nt;
        m2_strip(g.NT, 4 * rho + w4, &kb0, &nt);
        int lane = lane0, LD = g.ld;
        asm volatile("" : "+v"(lane), "+s"(LD));
        const u32 *tb0 = frag_lane_base(T0, g, lane), *tb1 = frag_lane_base(T1, g, lane);
        const unsigned char *st0 = stR + buf * 32 * g.pitchA + (lane & 31) * g.pitchA + 16 * (lane >> 5);
        u32 mlow[4];
        diag_low_mask(lane, mlow);
        auto epi = [&](auto &lo, auto &hi) {
          __syncthreads();                                           // the io wave has drained this chunk
          m2_dump(chunk, cp, LD, lane, lo, hi);
        };
        switch (nt) {
          case 0: __syncthreads(); break;
          case 1: toeplitz_strip<M_ENC, 1>(st0, st0, tb0, tb1, g, kb0, mlow, epi); break;
          case 2: toeplitz_strip<M_ENC, 2>(st0, st0, tb0, tb1, g, kb0, mlow, epi); break;
          case 3: toeplitz_strip<M_ENC, 3>(st0, st0, tb0, tb1, g, kb0, mlow, epi); break;
          default: toeplitz_strip<M_ENC, 4>(st0, st0, tb0, tb1, g, kb0, mlow, epi); break;
        }
        __syncthreads();                                             // chunk full; (last round) next stage complete
      }
    }
    return;
  }
  long drain_rb = -1; int drain_round = 0;                           // what the chunk holds
  M2EncPre pre;                                                      // the plaintext bytes of that chunk's rows
  auto drain = [&]() {
#if !(NTRU_ABLATE & 1024)
    if (drain_rb < 0) return;
    int dk, dn;
    m2_strip(g.NT, 4 * drain_round + w4, &dk, &dn);
    if (want_q) m2_drain_encrypt<true>(chunk, cp, dk, dn, drain_rb << 5, B, g.N, g.ld, q, pre, e, quotE, lane0);
    else m2_drain_encrypt<false>(chunk, cp, dk, dn, drain_rb << 5, B, g.N, g.ld, q, pre, e, quotE, lane0);
#endif
  };
  int it = 0;
  for (long rb = blockIdx.x; rb < nrb; rb += gridDim.x, it++) {
    const int buf = it & 1;
    for (int rho = 0; rho < rounds; rho++) {
      int kb0, nt;
      m2_strip(g.NT, 4 * rho + w4, &kb0, &nt);
      const bool stage_next = rho == 0 && rb + gridDim.x < nrb;
      if (stage_next) stage_load(rb + gridDim.x);                    // in flight during the drain
      drain();
      if (stage_next) stage_store(rb + gridDim.x, buf ^ 1);
      __syncthreads();                                               // chunk free for the dump
      drain_rb = rb; drain_round = rho;
      m2_prefetch_encrypt(pre, kb0, nt, rb << 5, B, g.N, g.ld, m, lane0);   // in flight while the matrix wave dumps
      __syncthreads();                                               // chunk full
    }
  }
  drain();
}

// ---- family 4 for PER-ITEM operands: verifyKeysInputs (index.js:141-197) on the matrix cores ---------------------
// No matrix is shared by the batch, but one product c = a * s is itself a 32-row matrix product per tile distance
// d = kb - ib (tools/peritem_mfma_model.py): C[kb][k'] += sum_i' F[kb - d][i'] G_d[i'][k'] with F the 32-coefficient chunks
// of a (rows = output tiles, read as aligned 16-byte pieces of a zero-padded natural-order byte array) and G_d the
// Toeplitz tile of s (fragments of the reversed cyclic array, as above).  One accumulator pair (low / high) holds the whole
// product of an item; a 13-bit operand contributes two digit planes with SEPARATE accumulators (value = acc0 + 128 acc1),
// so nothing is scaled.  2 NT - 1 (+1 for the split diagonal) matrix instructions per plane.  One item per wave, all LDS
// regions private to the wave, no workgroup barrier.
constexpr int PI_PAD = 32;          // zero chunks on either side of the chunk matrix: rows 0..31, distances +-(NT-1)
constexpr int PI_WAVES = 2;         // waves per workgroup (LDS, not registers, bounds the residency: ~15 KB per wave)
struct PGeom { int N, NT, tpitch; };
static __host__ __device__ inline size_t pi_fa_bytes(const PGeom &g) { return (size_t)32 * (g.NT + 2 * PI_PAD); }
static __host__ __device__ inline size_t pi_nat_bytes(const PGeom &g) { return ((size_t)3 * g.N + 64 + 15) & ~(size_t)15; }
// per wave: the two chunk matrices, then the reversed array.  The three natural-order periods the array is built from
// are staged OVER the chunk matrices (2 fa >= nat for every N <= 1024) and wiped again before the digits go in.
static __host__ __device__ inline size_t pi_wave_bytes(const PGeom &g) { return 2 * pi_fa_bytes(g) + (size_t)16 * g.tpitch; }

// Reversed cyclic array (4 byte-shifted copies) of the 16 bytes per lane in sv (coefficients 16 lane .. 16 lane + 15 of a
// ternary operand, zero at and beyond N): three periods in natural order (period k starts at byte k N, any alignment:
// unaligned LDS stores), then T[c][w] = bytes rev[4w + c + j], rev[y] = s[(Y0 - y) mod N], as byte-swapped unaligned reads.
static __device__ __forceinline__ void pi_build_array(unsigned char *nat, u32 *T, const PGeom &g, int lane, v4i sv) {
  const int N = g.N, Y0 = 32 * g.NT - 1;
  if (16 * lane < N) {
    union { v4i v; unsigned char c[16]; } u; u.v = sv;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (16 * lane + 16 <= N) *(v4i *)(nat + k * N + 16 * lane) = sv;
      else for (int j = 0; j < 16; j++) if (16 * lane + j < N) nat[k * N + 16 * lane + j] = u.c[j];
    }
    if (lane < 4) *(v4i *)(nat + 3 * N + 16 * lane) = sv;                 // N >= 64
  }
  wave_lds_fence();
  // Word w of copy c holds bytes nat[A .. A+3] reversed, A = E - c, E = Y0 + 2N - 3 - 4w.  E & 3 is the same for every
  // lane, so the four copies of a word come from three ALIGNED dwords around E >> 2 with one byte permute each
  // (an unaligned LDS dword read costs several aligned ones: the build was 28 % of a product in the probe).
  const u32 *D = (const u32 *)nat;
  const int e = __builtin_amdgcn_readfirstlane((Y0 + 2 * N - 3) & 3);
  u32 sel[4]; int dk[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int al = c <= e ? e - c : e - c + 4;                            // byte offset of A inside its dword
    dk[c] = c <= e ? 0 : -1;                                              // ... which is dword K or K - 1
    sel[c] = 0x00010203u + 0x01010101u * (u32)al;                         // bytes al+3, al+2, al+1, al of the pair (reversed)
  }
  for (int w = lane; w < g.tpitch; w += 64) {
    int K = (Y0 + 2 * N - 3 - 4 * w) >> 2;
    K = K < 1 ? 1 : K;                                                    // pad words of a copy are never read
    const u32 dm = D[K - 1], d0 = D[K], dp = D[K + 1];
#pragma unroll
    for (int c = 0; c < 4; c++)
      T[c * g.tpitch + w] = dk[c] == 0 ? __builtin_amdgcn_perm(dp, d0, sel[c]) : __builtin_amdgcn_perm(d0, dm, sel[c]);
  }
  wave_lds_fence();
  for (int i = 16 * lane; i < (int)pi_nat_bytes(g); i += 16 * 64) *(v4i *)(nat + i) = (v4i){0, 0, 0, 0};   // nat lies over the
  wave_lds_fence();                                                       // chunk matrices: their pads are zero again
}

// Digit planes of 16 values (u16 pairs in x[8], element i0 + j; zero at and beyond N) -> natural-order int8 bytes, on
// packed 16-bit pairs.  mul: the operand is (mul v) mod q (p fq of index.js:155; 1 otherwise).  q > 256: v = d0 + 128 d1 with
// d0 = v & 127, d1 = v >> 7 <= 63 (the two planes have SEPARATE accumulators, so nothing needs a signed representative).
// q <= 256: ONE plane, the centred representative in [-q/2, q/2) (d1 = 0; the callers skip that plane's matrix instructions).
static __device__ __forceinline__ void pi_digits(const u32 (&x)[8], u32 q, u32 mul, int i0, int N, v4i &o0, v4i &o1) {
  const u32 qm2 = (q - 1) * 0x00010001u;
  u32 v[8];
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const int left = N - (i0 + 2 * c);                     // valid elements of this pair
    const u32 keep = left >= 2 ? 0xFFFFFFFFu : (left == 1 ? 0x0000FFFFu : 0u);
    const u32 t = mul == 1u ? x[c] : as_u32(as_pair(x[c]) * (u16x2){(u16)mul, (u16)mul});
    v[c] = t & qm2 & keep;
  }
  if (q <= 256) {
    const u32 h2 = (q >> 1) * 0x00010001u;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const u32 a = as_u32(as_pair((as_u32(as_pair(v[2 * c]) + as_pair(h2)) & qm2)) - as_pair(h2));           // two's complement low bytes
      const u32 b = as_u32(as_pair((as_u32(as_pair(v[2 * c + 1]) + as_pair(h2)) & qm2)) - as_pair(h2));
      o0[c] = (int)__builtin_amdgcn_perm(b, a, 0x06040200u);
      o1[c] = 0;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const u32 a = v[2 * c], b = v[2 * c + 1];
      o0[c] = (int)__builtin_amdgcn_perm(b & 0x007F007Fu, a & 0x007F007Fu, 0x06040200u);
      o1[c] = (int)__builtin_amdgcn_perm((b >> 7) & 0x007F007Fu, (a >> 7) & 0x007F007Fu, 0x06040200u);
    }
  }
}

// One plane-pair product: acc{L,H}{0,1} += chunk matrices fa0 / fa1 (x) Toeplitz fragments of T.  TWO = false: one plane.
// A step (tile distance d) is one or two matrix instructions on operands that are used once, so the loop lives on its
// LDS reads: they are requested TWO steps ahead into three rotating register sets (unrolled by three, no register moves;
// one step ahead left the wave waiting on LDS latency at the top of every iteration: 64 matrix clocks per step against
// ~130 of latency).  The first instruction of every accumulator takes C = 0.
template <bool TWO>
static __device__ __forceinline__ void pi_product(const unsigned char *pa0, const unsigned char *pa1, const u32 *tb, int NT,
                                                  const u32 (&mlow)[4], v16i &L0, v16i &L1, v16i &H0, v16i &H1) {
  struct Ops { v4i a0, a1, w; };
  auto ld = [&](int d, Ops &o) {                          // distance d: fragment 8 d dwords below the lane's base, rows shifted by d
    d = d < NT ? d : NT;                                  // requests past the last step read pad bytes (never used)
    const u32 *p = tb - 8 * d;
    o.w = (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
    o.a0 = *(const v4i *)(pa0 - 32 * d);
    if (TWO) o.a1 = *(const v4i *)(pa1 - 32 * d);
  };
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto step = [&](int d, const Ops &o) {                  // d < 0: high, d > 0: low, d == 0: split by the diagonal mask
    if (d < 0) {
      H0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a0, o.w, H0, 0, 0, 0);
      if (TWO) H1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a1, o.w, H1, 0, 0, 0);
    } else if (d > 0) {
      L0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a0, o.w, L0, 0, 0, 0);
      if (TWO) L1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a1, o.w, L1, 0, 0, 0);
    } else {
      const v4i wl = and4(o.w, mlow);
      const v4i wh = {(int)((u32)o.w[0] & ~mlow[0]), (int)((u32)o.w[1] & ~mlow[1]), (int)((u32)o.w[2] & ~mlow[2]), (int)((u32)o.w[3] & ~mlow[3])};
      L0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a0, wl, zero, 0, 0, 0);          // the first term of `low`
      H0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a0, wh, H0, 0, 0, 0);
      if (TWO) {
        L1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a1, wl, zero, 0, 0, 0);
        H1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(o.a1, wh, H1, 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int i = 0; i < 16; i++) { H0[i] = 0; H1[i] = 0; }   // NT = 1 has no d < 0 step; otherwise folded into the first step below
  Ops A, Bq, C;
  int d = -(NT - 1);
  ld(d, A); ld(d + 1, Bq);
  // high part: steps d = -(NT-1) .. -1, three per trip
  for (; d + 2 < 0; d += 3) {
    ld(d + 2, C); step(-1, A);
    ld(d + 3, A); step(-1, Bq);
    ld(d + 4, Bq); step(-1, C);
  }
  // 0, 1 or 2 steps of the high part are left; then the diagonal; then the low part.  The rotation continues with moves
  // for these few steps (at most two high steps + the diagonal), after which the low part runs three per trip again.
  for (; d < 0; d++) {
    ld(d + 2, C); step(-1, A);
    A = Bq; Bq = C;
  }
  ld(2, C); step(0, A);                                   // d == 0
  A = Bq; Bq = C;
  d = 1;
  for (; d + 2 < NT; d += 3) {
    ld(d + 2, C); step(1, A);
    ld(d + 3, A); step(1, Bq);
    ld(d + 4, Bq); step(1, C);
  }
  for (; d < NT; d++) {
    ld(d + 2, C); step(1, A);
    A = Bq; Bq = C;
  }
  if (!TWO) {
#pragma unroll
    for (int i = 0; i < 16; i++) { L1[i] = 0; H1[i] = 0; }
  }
}

static __device__ __forceinline__ int wave_max(int v) {
  // the lane index is re-materialised here: otherwise the six permute addresses are hoisted out of the item loop and,
  // at the register limit of three waves per SIMD, spilled to scratch
  int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int o = __builtin_amdgcn_ds_bpermute((l ^ off) << 2, v);
    v = o > v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_verify_keys_m(
    PGeom g, u32 q, const int8_t *__restrict__ f, const int8_t *__restrict__ gg, const u16 *__restrict__ fq,
    const uint8_t *__restrict__ fp, const u16 *__restrict__ h, long B, u16 *__restrict__ quot_fq,
    u16 *__restrict__ rem_fq, uint8_t *__restrict__ quot_fp, uint8_t *__restrict__ rem_fp, u16 *__restrict__ quot_h,
    u16 *__restrict__ rem_h, uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
  unsigned char *fa0 = lds + (size_t)wave * pi_wave_bytes(g), *fa1 = fa0 + pi_fa_bytes(g), *nat = fa0;
  u32 *T = (u32 *)(fa1 + pi_fa_bytes(g));
  const int N = g.N, NT = g.NT;
  for (size_t i = 16 * lane; i < 2 * pi_fa_bytes(g); i += 16 * 64) *(v4i *)(fa0 + i) = (v4i){0, 0, 0, 0};   // the pads stay zero
  const int y0 = 32 * NT - 1 - r + 16 * hh;
  const u32 *tb = T + (y0 & 3) * g.tpitch + (y0 >> 2);
  const unsigned char *pa0 = fa0 + 32 * PI_PAD + 32 * r + 16 * hh, *pa1 = fa1 + 32 * PI_PAD + 32 * r + 16 * hh;
  u32 mlow[4];
  diag_low_mask(lane, mlow);
  const bool stager = 16 * lane < 32 * NT;                                 // lanes that hold a 16-coefficient chunk of a row
  const v4i cmask = col_mask16(16 * lane, N);                              // bytes of this lane's chunk that are below N
  const int kl = 128 * hh + r;                                             // accumulator register i holds index 32 ((i&3) + 8 (i>>2)) + kl
  wave_lds_fence();
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += (long)gridDim.x * PI_WAVES) {
    const long row = item * N, left = (B - item) * N;
    u32 fl = 0;
    // Every operand row is requested one product ahead of its use (rows at any alignment: aligned chunks + a wave-uniform
    // byte shift at use): fq, f and fp at the top, g and fq again (an L2 hit) before product 2, h before product 3.  A fetch
    // right where each product needs it left the wave idle for a round trip to HBM three times per item.
    const AlignedSrc s_fq = aligned_src(fq + row, 2 * left), s_f = aligned_src(f + row, left), s_g = aligned_src(gg + row, left),
                     s_fp = aligned_src(fp + row, left);
    RawChunks<2> r_fq = load_raw<2>(s_fq, s_fq.a0 + 32 * lane, 0);
    const RawChunks<1> r_f = load_raw<1>(s_f, s_f.a0 + 16 * lane, 0);
    const RawChunks<1> r_fp = load_raw<1>(s_fp, s_fp.a0 + 16 * lane, 0);
    auto bytes_of = [&](const RawChunks<1> &rw, const AlignedSrc &sr) {
      v4i v[1];
      shift_raw<1>(rw, __builtin_amdgcn_readfirstlane(sr.a0), v);
      return v[0];
    };
    auto fq_pairs = [&](u32 (&x)[8]) {                                    // 16 coefficients per lane as u16 pairs
      v4i v[2];
      shift_raw<2>(r_fq, __builtin_amdgcn_readfirstlane(s_fq.a0), v);
#pragma unroll
      for (int c = 0; c < 4; c++) { x[c] = (u32)v[0][c]; x[4 + c] = (u32)v[1][c]; }
    };
    // ternary operands: any negative byte is -1 (ValTernary), bytes at and beyond N are zero -- four bytes at a time
    auto ternary = [&](v4i v) {
      v4i o;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const u32 w = (u32)(v[c] & cmask[c]);
        u32 neg = (w >> 7) & 0x01010101u;                                  // 1 in every negative byte ...
        neg |= neg << 1; neg |= neg << 2; neg |= neg << 4;                 // ... spread to 0xFF
        o[c] = (int)(w | neg);
      }
      return o;
    };
    // ---- product 1: fq * f mod q (index.js:158-160)
    {
      u32 xq[8];
      fq_pairs(xq);
      const v4i tf = ternary(bytes_of(r_f, s_f));
      pi_build_array(nat, T, g, lane, tf);
      if (stager) {
        v4i o0, o1;
        pi_digits(xq, q, 1u, 16 * lane, N, o0, o1);
        *(v4i *)(fa0 + 32 * PI_PAD + 16 * lane) = o0;
        *(v4i *)(fa1 + 32 * PI_PAD + 16 * lane) = o1;
      }
      wave_lds_fence();
    }
    v16i L0, L1, H0, H1;
    pi_product<true>(pa0, pa1, tb, NT, mlow, L0, L1, H0, H1);
    {
      // stores through one-row descriptors: index k = 32 kb + r is a per-lane offset (128 hh + r) plus a compile-time
      // one per register, indices >= N fall outside the descriptor and are dropped -- no address arithmetic per store
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_fq + row, 2L * N), rs_q = rows_rsrc(quot_fq + row, 2L * N);
      bool nz_hi = false, first_not_one = false;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2)), k = ko + kl;
        const int lo = L0[i] + 128 * L1[i], hi = H0[i] + 128 * H1[i];
        const u32 rv = (u32)(lo + hi) & (q - 1);
        __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
        nz_hi |= k >= 1 && k < N && rv != 0;
        first_not_one |= k == 0 && rv != 1;
      }
      if (__ballot(nz_hi) != 0 && __ballot(first_not_one) != 0) fl |= NTRU_FLAG_INVALID_FQ;   // length !== 1 && [0] !== 1
    }
    wave_lds_fence();
    // ---- product 2: fp * f mod p (index.js:161-163): the array of f serves again, one plane of fp mod 3
    {
      const v4i vfp = bytes_of(r_fp, s_fp);
      union { v4i v; unsigned char c[16]; } u; u.v = vfp & cmask;
#pragma unroll
      for (int j = 0; j < 16; j++) u.c[j] = (unsigned char)((u32)u.c[j] % 3u);
      if (stager) *(v4i *)(fa0 + 32 * PI_PAD + 16 * lane) = u.v;
    }
    wave_lds_fence();
    r_fq = load_raw<2>(s_fq, s_fq.a0 + 32 * lane, 0);                       // for product 3, in flight during product 2
    const RawChunks<1> r_g = load_raw<1>(s_g, s_g.a0 + 16 * lane, 0);
    pi_product<false>(pa0, pa1, tb, NT, mlow, L0, L1, H0, H1);
    {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_fp + row, (long)N), rs_q = rows_rsrc(quot_fp + row, (long)N);
      bool nz_hi = false, first_not_one = false;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2)), k = ko + kl;
        // |L + H|, |H| <= 127 N (f is an int8, fp < 3): a multiple of 3 above that keeps the dividend non-negative
        const u32 x = (u32)(L0[i] + H0[i] + 3 * 131072), y = (u32)(3 * 131072 - H0[i]);
        const u32 rv = x % 3u, qv = y % 3u;
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)rv, rs_r, kl, ko, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)qv, rs_q, kl, ko, 0);
        nz_hi |= k >= 1 && k < N && rv != 0;
        first_not_one |= k == 0 && rv != 1;
      }
      if (__ballot(nz_hi) != 0 && __ballot(first_not_one) != 0) fl |= NTRU_FLAG_INVALID_FP;
    }
    wave_lds_fence();
    // ---- product 3: ((p fq) mod q) * g mod q, compared with h below its trimmed length (index.js:155,164-166)
    {
      u32 xq[8];
      fq_pairs(xq);
      const v4i tg = ternary(bytes_of(r_g, s_g));
      pi_build_array(nat, T, g, lane, tg);
      if (stager) {
        v4i o0, o1;
        pi_digits(xq, q, 3u, 16 * lane, N, o0, o1);
        *(v4i *)(fa0 + 32 * PI_PAD + 16 * lane) = o0;
        *(v4i *)(fa1 + 32 * PI_PAD + 16 * lane) = o1;
      }
      wave_lds_fence();
    }
    // h is requested before the product whose remainder it is compared with, as a row chunk (16 coefficients per lane);
    // the remainder gets into the same layout through the wave's LDS (the natural-order area is free again by then)
    const AlignedSrc s_h = aligned_src(h + row, 2 * left);
    const RawChunks<2> r_h = load_raw<2>(s_h, s_h.a0 + 32 * lane, 0);
    pi_product<true>(pa0, pa1, tb, NT, mlow, L0, L1, H0, H1);
    {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_h + row, 2L * N), rs_q = rows_rsrc(quot_h + row, 2L * N);
      u16 *remx = (u16 *)nat;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const int lo = L0[i] + 128 * L1[i], hi = H0[i] + 128 * H1[i];
        const u32 rv = (u32)(lo + hi) & (q - 1);
        __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
        remx[ko + kl] = (u16)rv;                                          // ko + kl < 32 NT <= (3 N + 64) / 2
      }
      wave_lds_fence();
      // index.js:165: h[k] must equal the remainder for every k below h's trimmed length
      v4i hc[2];
      shift_raw<2>(r_h, __builtin_amdgcn_readfirstlane(s_h.a0), hc);
      const int i0 = 16 * lane;
      u32 nz = 0, df = 0;                                                 // bit j: h[i0 + j] != 0 / != remainder[i0 + j]
      if (stager) {
        const v4i rc0 = *(const v4i *)(nat + 32 * lane), rc1 = *(const v4i *)(nat + 32 * lane + 16);
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const int lf = N - (i0 + 2 * c);
          const u32 keep = lf >= 2 ? 0xFFFFFFFFu : (lf == 1 ? 0x0000FFFFu : 0u);
          const u32 hx = (u32)(c < 4 ? hc[0][c] : hc[1][c - 4]) & keep, rx = (u32)(c < 4 ? rc0[c] : rc1[c - 4]) & keep;
          const u32 x = hx ^ rx;
          nz |= ((hx & 0xFFFFu) ? 1u : 0u) << (2 * c) | ((hx >> 16) ? 2u : 0u) << (2 * c);
          df |= ((x & 0xFFFFu) ? 1u : 0u) << (2 * c) | ((x >> 16) ? 2u : 0u) << (2 * c);
        }
      }
      const int top = nz ? i0 + 31 - __builtin_clz(nz) : -1;
      const int wtop = wave_max(top);
      const int hl = wtop >= 0 ? wtop + 1 : 1;                          // trimmed length of h (1 for the zero polynomial)
      const int nv = hl - i0 < 0 ? 0 : (hl - i0 > 16 ? 16 : hl - i0);   // this lane's indices below hl
      if (__ballot((df & ((1u << nv) - 1u)) != 0) != 0) fl |= NTRU_FLAG_INVALID_H;
    }
    if (lane == 0) flags[item] = (uint8_t)fl;
    wave_lds_fence();
  }
}

// One per-item product on the matrix cores: rem (and quot) of ((mul a) mod q) * s split by 1 - x^N, a < 2^16 per item,
// s ternary per item: generatePublicKeyH (index.js:72-79, mul = p) and the f * t product of polyInv's Newton rounds
// (index.js:499-506, mul = 1).  Same machinery as k_verify_keys_m.
__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_product_tern_m(
    PGeom g, u32 q, u32 mul, const u16 *__restrict__ a, const int8_t *__restrict__ s, long B, u16 *__restrict__ quot,
    u16 *__restrict__ rem) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
  unsigned char *fa0 = lds + (size_t)wave * pi_wave_bytes(g), *fa1 = fa0 + pi_fa_bytes(g), *nat = fa0;
  u32 *T = (u32 *)(fa1 + pi_fa_bytes(g));
  const int N = g.N, NT = g.NT;
  for (size_t i = 16 * lane; i < 2 * pi_fa_bytes(g); i += 16 * 64) *(v4i *)(fa0 + i) = (v4i){0, 0, 0, 0};   // the pads stay zero
  const int y0 = 32 * NT - 1 - r + 16 * hh;
  const u32 *tb = T + (y0 & 3) * g.tpitch + (y0 >> 2);
  const unsigned char *pa0 = fa0 + 32 * PI_PAD + 32 * r + 16 * hh, *pa1 = fa1 + 32 * PI_PAD + 32 * r + 16 * hh;
  u32 mlow[4];
  diag_low_mask(lane, mlow);
  const bool stager = 16 * lane < 32 * NT;
  const v4i cmask = col_mask16(16 * lane, N);
  const bool want_q = quot != nullptr;
  wave_lds_fence();
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += (long)gridDim.x * PI_WAVES) {
    const long row = item * N, left = (B - item) * N;
    {
      const AlignedSrc sa = aligned_src(a + row, 2 * left), ss = aligned_src(s + row, left);
      const RawChunks<2> ra = load_raw<2>(sa, sa.a0 + 32 * lane, 0);
      const RawChunks<1> rs = load_raw<1>(ss, ss.a0 + 16 * lane, 0);
      v4i va[2], vs[1];
      shift_raw<2>(ra, __builtin_amdgcn_readfirstlane(sa.a0), va);
      shift_raw<1>(rs, __builtin_amdgcn_readfirstlane(ss.a0), vs);
      u32 xa[8];
#pragma unroll
      for (int c = 0; c < 4; c++) { xa[c] = (u32)va[0][c]; xa[4 + c] = (u32)va[1][c]; }
      union { v4i v; signed char c[16]; } u; u.v = vs[0] & cmask;        // any negative byte is -1 (ValTernary)
#pragma unroll
      for (int j = 0; j < 16; j++) u.c[j] = u.c[j] < 0 ? (signed char)-1 : u.c[j];
      pi_build_array(nat, T, g, lane, u.v);
      if (stager) {
        v4i o0, o1;
        pi_digits(xa, q, mul, 16 * lane, N, o0, o1);
        *(v4i *)(fa0 + 32 * PI_PAD + 16 * lane) = o0;
        *(v4i *)(fa1 + 32 * PI_PAD + 16 * lane) = o1;
      }
      wave_lds_fence();
    }
    v16i L0, L1, H0, H1;
    if (q <= 256) pi_product<false>(pa0, pa1, tb, NT, mlow, L0, L1, H0, H1);   // one digit plane (early Newton rounds, small q)
    else pi_product<true>(pa0, pa1, tb, NT, mlow, L0, L1, H0, H1);
    {
      const int kl = 128 * hh + r;                                       // see k_verify_keys_m: indices >= N are dropped
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem + row, 2L * N);
      const __amdgpu_buffer_rsrc_t rs_q = rows_rsrc(want_q ? quot + row : nullptr, want_q ? 2L * N : 0L);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const int lo = L0[i] + 128 * L1[i], hi = H0[i] + 128 * H1[i];
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(lo + hi) & (q - 1)), rs_r, 2 * kl, 2 * ko, 0);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
      }
    }
    wave_lds_fence();
  }
}

// Generic per-item product on the matrix cores: both operands < q <= 8192 (multiplyPolynomials + dividePolynomials by I,
// index.js:319-401, with q a power of two; the v * v product of polyInv's Newton rounds).  With a = a0 + 128 a1 and
// b = b0 + 128 b1 the product is a0 b0 + 128 (a0 b1 + a1 b0) + 16384 a1 b1, and 16384 = 0 mod q: three plane products, two
// accumulator groups, two reversed arrays (the digit planes of b) per item.
static __host__ __device__ inline size_t pi_wave_bytes2(const PGeom &g) { return pi_wave_bytes(g) + (size_t)16 * g.tpitch; }

__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_polymul_m(
    PGeom g, u32 q, const u16 *__restrict__ a, const u16 *__restrict__ b, long B, u16 *__restrict__ quot,
    u16 *__restrict__ rem) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
  unsigned char *fa0 = lds + (size_t)wave * pi_wave_bytes2(g), *fa1 = fa0 + pi_fa_bytes(g), *nat = fa0;
  u32 *T0 = (u32 *)(fa1 + pi_fa_bytes(g)), *T1 = T0 + 4 * g.tpitch;
  const int N = g.N, NT = g.NT;
  for (size_t i = 16 * lane; i < 2 * pi_fa_bytes(g); i += 16 * 64) *(v4i *)(fa0 + i) = (v4i){0, 0, 0, 0};   // the pads stay zero
  const int y0 = 32 * NT - 1 - r + 16 * hh;
  const u32 *tb0 = T0 + (y0 & 3) * g.tpitch + (y0 >> 2), *tb1 = tb0 + 4 * g.tpitch;
  const unsigned char *pa0 = fa0 + 32 * PI_PAD + 32 * r + 16 * hh, *pa1 = fa1 + 32 * PI_PAD + 32 * r + 16 * hh;
  u32 mlow[4];
  diag_low_mask(lane, mlow);
  const bool stager = 16 * lane < 32 * NT;
  const bool one = q <= 256;                               // single int8 plane per operand (pi_digits)
  wave_lds_fence();
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += (long)gridDim.x * PI_WAVES) {
    const long row = item * N, left = (B - item) * N;
    {
      auto fetch = [&](const u16 *base, u32 (&x)[8]) {
        const AlignedSrc sr = aligned_src(base + row, 2 * left);
        const RawChunks<2> rw = load_raw<2>(sr, sr.a0 + 32 * lane, 0);
        v4i v[2];
        shift_raw<2>(rw, __builtin_amdgcn_readfirstlane(sr.a0), v);
#pragma unroll
        for (int c = 0; c < 4; c++) { x[c] = (u32)v[0][c]; x[4 + c] = (u32)v[1][c]; }
      };
      u32 xa[8], xb[8];
      fetch(a, xa); fetch(b, xb);
      v4i a0, a1, b0, b1;
      pi_digits(xa, q, 1u, 16 * lane, N, a0, a1);
      pi_digits(xb, q, 1u, 16 * lane, N, b0, b1);
      pi_build_array(nat, T0, g, lane, b0);
      if (!one) pi_build_array(nat, T1, g, lane, b1);
      if (stager) {
        *(v4i *)(fa0 + 32 * PI_PAD + 16 * lane) = a0;
        *(v4i *)(fa1 + 32 * PI_PAD + 16 * lane) = a1;
      }
      wave_lds_fence();
    }
    v16i L0, L1, H0, H1;                                   // group 0: a0 b0; group 1: a0 b1 + a1 b0
#pragma unroll
    for (int i = 0; i < 16; i++) { L0[i] = 0; L1[i] = 0; H0[i] = 0; H1[i] = 0; }
    auto ld = [&](int d, v4i &x0, v4i &x1, v4i &w0, v4i &w1) {
      const u32 *p0 = tb0 - 8 * d, *p1 = tb1 - 8 * d;
      w0 = (v4i){(int)p0[0], (int)p0[1], (int)p0[2], (int)p0[3]};
      w1 = (v4i){(int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
      x0 = *(const v4i *)(pa0 - 32 * d);
      x1 = *(const v4i *)(pa1 - 32 * d);
    };
    auto mm3 = [&](v16i &X0, v16i &X1, v4i x0, v4i x1, v4i w0, v4i w1) {
      X0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x0, w0, X0, 0, 0, 0);
      if (!one) {                                          // q <= 256: both operands are single planes
        X1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x0, w1, X1, 0, 0, 0);
        X1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x1, w0, X1, 0, 0, 0);
      }
    };
    v4i x0, x1, w0, w1;
    ld(-(NT - 1), x0, x1, w0, w1);
    for (int d = -(NT - 1); d < 0; d++) {
      v4i n0, n1, m0, m1;
      ld(d + 1, n0, n1, m0, m1);
      mm3(H0, H1, x0, x1, w0, w1);
      x0 = n0; x1 = n1; w0 = m0; w1 = m1;
    }
    {
      v4i n0, n1, m0, m1;
      ld(1, n0, n1, m0, m1);
      u32 mhigh[4];
#pragma unroll
      for (int c = 0; c < 4; c++) mhigh[c] = ~mlow[c];
      mm3(L0, L1, x0, x1, and4(w0, mlow), and4(w1, mlow));
      mm3(H0, H1, x0, x1, and4(w0, mhigh), and4(w1, mhigh));
      x0 = n0; x1 = n1; w0 = m0; w1 = m1;
    }
    for (int d = 1; d < NT; d++) {
      v4i n0, n1, m0, m1;
      ld(d + 1, n0, n1, m0, m1);
      mm3(L0, L1, x0, x1, w0, w1);
      x0 = n0; x1 = n1; w0 = m0; w1 = m1;
    }
    {
      const int kl = 128 * hh + r;                                       // see k_verify_keys_m: indices >= N are dropped
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem + row, 2L * N), rs_q = rows_rsrc(quot + row, 2L * N);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const u32 lo = (u32)L0[i] + 128u * (u32)L1[i], hi = (u32)H0[i] + 128u * (u32)H1[i];
        __builtin_amdgcn_raw_buffer_store_b16((u16)((lo + hi) & (q - 1)), rs_r, 2 * kl, 2 * ko, 0);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((0u - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
      }
    }
    wave_lds_fence();
  }
}

// ---- host side ----------------------------------------------------------------------------------------------

static thread_local std::string g_err;
int ntru_fail(int code, const std::string &msg) { g_err = msg; return code; }
static inline int fail(int code, const std::string &msg) { return ntru_fail(code, msg); }

int ntru_grow_dev(GrowBuf *b, size_t bytes) {
  if (bytes <= b->cap) return NTRU_OK;
  if (b->p) (void)hipFree(b->p);          // waits for the device: nothing in flight still reads the old buffer
  b->p = nullptr; b->cap = 0;
  const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  if (hipMalloc(&b->p, want) != hipSuccess) { b->p = nullptr; return fail(NTRU_ERR_HIP, "hipMalloc failed"); }
  b->cap = want;
  return NTRU_OK;
}

int ntru_grow_pinned(GrowBuf *b, size_t bytes) {
  if (bytes <= b->cap) return NTRU_OK;
  if (b->p) (void)hipHostFree(b->p);
  b->p = nullptr; b->cap = 0;
  const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  if (hipHostMalloc(&b->p, want, hipHostMallocDefault) != hipSuccess) { b->p = nullptr; return fail(NTRU_ERR_HIP, "hipHostMalloc failed"); }
  b->cap = want;
  return NTRU_OK;
}

// Occupancy of (kernel, block size, LDS bytes), asked once per engine; the first use of a kernel with more than 64 KiB
// of dynamic LDS also raises its limit.
int ntru_blocks_per_cu(ntru_engine *eng, const void *fn, int threads, size_t lds, int *per_cu) {
  for (int i = 0; i < eng->n_occ; i++)
    if (eng->occ[i].fn == fn && eng->occ[i].lds == lds && eng->occ[i].threads == threads) { *per_cu = eng->occ[i].per_cu; return NTRU_OK; }
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int n = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds));
  if (n < 1) n = 1;
  if (eng->n_occ < (int)(sizeof eng->occ / sizeof eng->occ[0])) eng->occ[eng->n_occ++] = {fn, lds, threads, n};
  *per_cu = n;
  return NTRU_OK;
}

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static int pick_K(int N) {
  for (int K = 1; K <= 15; K += 2)
    if ((N + 2 * K - 1) / (2 * K) <= 64) return K;
  return 0;
}

extern "C" int ntru_engine_supports(int N, int mod) {
  if (N < 2 || N > NTRU_MAX_N || mod < 2) return 0;
  if (is_pow2(mod)) return mod <= 65536;
  return (long)N * (mod - 1) * (mod - 1) < 65536;
}

static Geom make_geom(int N, int K) {
  Geom g;
  g.N = N;
  g.nl = (N + 2 * K - 1) / (2 * K);
  g.G = 64 / g.nl;
  g.off = K * g.nl;
  g.eo_len = 2 * K * g.nl;
  g.a_len = K * g.nl;
  return g;
}

extern "C" int ntru_engine_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char *ntru_last_error(void) { return g_err.c_str(); }

extern "C" int ntru_engine_create(int device, ntru_engine_t **out) {
  if (!out) return fail(NTRU_ERR_ARG, "ntru_engine_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(NTRU_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(e) +
                                        "); this engine has no CPU fallback");
  if (device < 0 || device >= n) return fail(NTRU_ERR_NO_DEVICE, "device id out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  ntru_engine *eng = new ntru_engine();
  eng->device = device;
  eng->stream = nullptr;
  eng->cus = prop.multiProcessorCount;
  eng->path = 0;
  eng->last_kernel[0] = 0;
  eng->n_occ = 0;
  eng->cur_scratch = &eng->scratch_dev;
  eng->max_blocks_per_cu = 0;
  if (const char *cap = getenv("NTRU_MAX_BLOCKS_PER_CU")) {      // tuning experiments only; read once
    const int c = atoi(cap);
    if (c >= 1) eng->max_blocks_per_cu = c;
  }
  *out = eng;
  return NTRU_OK;
}

extern "C" void ntru_engine_destroy(ntru_engine_t *eng) {
  if (!eng) return;
  (void)hipSetDevice(eng->device);
  for (HostSlot &s : eng->slot) {
    if (s.stream) { (void)hipStreamSynchronize(s.stream); (void)hipStreamDestroy(s.stream); }
    if (s.pinned.p) (void)hipHostFree(s.pinned.p);
    if (s.dev.p) (void)hipFree(s.dev.p);
    if (s.scratch.p) (void)hipFree(s.scratch.p);
  }
  if (eng->shared_dev.p) (void)hipFree(eng->shared_dev.p);
  if (eng->scratch_dev.p) (void)hipFree(eng->scratch_dev.p);
  delete eng;
}

extern "C" int ntru_engine_set_stream(ntru_engine_t *eng, void *hip_stream) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  eng->stream = (hipStream_t)hip_stream;
  return NTRU_OK;
}

extern "C" int ntru_engine_set_kernel_path(ntru_engine_t *eng, int path) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (path < 0 || path > 8)
    return fail(NTRU_ERR_ARG, "kernel path must be 0 (auto), 1 (MAC), 2 (add), 3 (add without dot8), 4 (matrix cores, two workgroups per CU), "
                              "5 (matrix cores, lock-step groups) or 6 (matrix cores, role-split encrypt)");
  eng->path = path;
  return NTRU_OK;
}

extern "C" const char *ntru_engine_last_kernel(ntru_engine_t *eng) { return eng ? eng->last_kernel : ""; }

static void note_kernel(ntru_engine *eng, const char *family, int K, int me) {
  if (me >= 0) snprintf(eng->last_kernel, sizeof eng->last_kernel, "%s<%d,%d>", family, K, me);
  else snprintf(eng->last_kernel, sizeof eng->last_kernel, "%s<%d>", family, K);
}

extern "C" int ntru_engine_synchronize(ntru_engine_t *eng) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  HIP_TRY(hipSetDevice(eng->device));
  HIP_TRY(hipStreamSynchronize(eng->stream));
  return NTRU_OK;
}

struct Launch { Geom g; int K; dim3 grid; size_t lds; };

// shared_eo: number of EO arrays shared by the workgroup; per_item_eo: whether each item also needs its own EO array.
static int plan(const ntru_engine *eng, int N, long B, int shared_eo, bool per_item_eo, Launch *L) {
  int K = pick_K(N);
  if (!K) return fail(NTRU_ERR_UNSUPPORTED, "N too large");
  L->K = K;
  L->g = make_geom(N, K);
  const size_t raw_len = ((size_t)N + 1) & ~(size_t)1;
  size_t per_wave = (size_t)L->g.G * ((size_t)L->g.a_len * 4 + (per_item_eo ? (size_t)L->g.eo_len * 8 + raw_len * 2 : 0));
  L->lds = (size_t)shared_eo * L->g.eo_len * 8 + WAVES_PER_BLOCK * per_wave;
  if (L->lds > 160 * 1024) return fail(NTRU_ERR_UNSUPPORTED, "parameter set needs more than 160 KiB of LDS");
  long ngroups = (B + L->g.G - 1) / L->g.G;
  long blocks = (ngroups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (blocks < 1) blocks = 1;                   // work blocks; capped to residency by resident_grid()
  L->grid = dim3((unsigned)blocks);
  return NTRU_OK;
}

// The add path needs one item per wave, 2K step bits in half a dword (K <= 7) and at least K values below q fitting
// a 16-bit field on top of a masked one.  Returns the mask interval ME (2K: once per block, K: twice) or 0.
static int add_path_me(const ntru_engine *eng, int N, int q) {
  if (eng->path == 1) return 0;
  int K = pick_K(N);
  if (!K || K > 7) return 0;
  Geom g = make_geom(N, K);
  if (g.G != 1) return 0;
  long limit = 65535 / (q - 1) - 1;          // additions of values < q allowed on a masked field
  if (limit >= 2 * K) return 2 * K;
  if (limit >= K) return K;
  return 0;
}

static int plan_add(const ntru_engine *eng, int N, long B, size_t shared_bytes, size_t per_wave_bytes, Launch *L) {
  L->K = pick_K(N);
  L->g = make_geom(N, L->K);
  L->lds = shared_bytes + WAVES_PER_BLOCK * per_wave_bytes;
  if (L->lds > 160 * 1024) return fail(NTRU_ERR_UNSUPPORTED, "parameter set needs more than 160 KiB of LDS");
  long blocks = (B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (blocks < 1) blocks = 1;
  L->grid = dim3((unsigned)blocks);
  return NTRU_OK;
}

// Shared-stepping decrypt: two items per wave, K pairs per lane.  Returns K (9/11/13) and the mask interval, or 0.
static int shared_path_K(const ntru_engine *eng, int N, int q, int p, int *me) {
  if (eng->path == 1 || p != 3 || (N & 1) == 0) return 0;
  int K = (N + 63) / 64;
  if ((K & 1) == 0) K++;
  if (K < 9 || K > 13) return 0;
  if ((long)N * 4 >= 65536) return 0;
  long limit = 65535 / (q - 1) - 1;
  if (limit >= K) *me = K; else if (limit >= 7) *me = 7; else return 0;
  if (K == 13 && *me == 7) return 0;      // that variant does not fit 128 VGPRs without spilling: use the MAC kernels
  return K;
}

// Matrix-core path (family 4): shared key, q a power of two <= 8192 (two int8 digit planes), LDS for a 32-row block.
static bool make_mgeom(const ntru_engine *eng, int N, int q, int ld, MGeom *g) {
  if (eng->path != 0 && eng->path < 4) return false;
  if (q > 8192 || N > 1024 || ld > 1024 || N < (eng->path >= 4 || ld != N ? 2 : 64)) return false;   // staging: lane = 16-byte chunk of a row
  g->N = N;
  g->ld = ld;
  g->NT = (N + 31) / 32;
  g->pitchA = 32 * g->NT + 16;
  g->tpitch = ((16 * g->NT + 31) / 32) * 32 + 8;
  return true;
}

#define DISPATCH_K_SHARED(Kv, MEv, D8v, ...)                                                        \
  switch ((Kv) * 1000 + (MEv) * 10 + (D8v)) {                                                       \
    case 13131: { constexpr int KK = 13, MM = 13; constexpr bool DD = true; __VA_ARGS__; } break;   \
    case 13130: { constexpr int KK = 13, MM = 13; constexpr bool DD = false; __VA_ARGS__; } break;  \
    case 11111: { constexpr int KK = 11, MM = 11; constexpr bool DD = true; __VA_ARGS__; } break;   \
    case 11110: { constexpr int KK = 11, MM = 11; constexpr bool DD = false; __VA_ARGS__; } break;  \
    case 11071: { constexpr int KK = 11, MM = 7; constexpr bool DD = true; __VA_ARGS__; } break;    \
    case 11070: { constexpr int KK = 11, MM = 7; constexpr bool DD = false; __VA_ARGS__; } break;   \
    case 9090: { constexpr int KK = 9, MM = 9; constexpr bool DD = false; __VA_ARGS__; } break;     \
    case 9070: { constexpr int KK = 9, MM = 7; constexpr bool DD = false; __VA_ARGS__; } break;     \
    default: return fail(NTRU_ERR_UNSUPPORTED, "no shared-step kernel for this (K, mask interval)"); \
  }

#define DISPATCH_K_ADD(Kv, MEv, ...)                                                                \
  switch ((Kv) * 100 + (MEv)) {                                                                     \
    case 714: { constexpr int KK = 7, MM = 14; __VA_ARGS__; } break;                                \
    case 707: { constexpr int KK = 7, MM = 7; __VA_ARGS__; } break;                                 \
    case 510: { constexpr int KK = 5, MM = 10; __VA_ARGS__; } break;                                \
    case 505: { constexpr int KK = 5, MM = 5; __VA_ARGS__; } break;                                 \
    case 306: { constexpr int KK = 3, MM = 6; __VA_ARGS__; } break;                                 \
    case 303: { constexpr int KK = 3, MM = 3; __VA_ARGS__; } break;                                 \
    case 102: { constexpr int KK = 1, MM = 2; __VA_ARGS__; } break;                                 \
    case 101: { constexpr int KK = 1, MM = 1; __VA_ARGS__; } break;                                 \
    default: return fail(NTRU_ERR_UNSUPPORTED, "no add-path kernel for this (K, mask interval)");   \
  }

// Persistent grid: as many workgroups as are co-resident (occupancy query for this kernel and LDS size) x CUs, capped
// by the work available.  A grid larger than residency would run its tail at a fraction of the chip.
template <class Kern>
static int resident_grid(const ntru_engine *eng, Kern kern, size_t lds, long work_blocks, dim3 *grid, int threads = BLOCK_THREADS) {
  int per_cu = 0;
  if (int rc = ntru_blocks_per_cu(const_cast<ntru_engine *>(eng), (const void *)kern, threads, lds, &per_cu)) return rc;
  if (eng->max_blocks_per_cu && eng->max_blocks_per_cu < per_cu) per_cu = eng->max_blocks_per_cu;
  long blocks = (long)eng->cus * per_cu;
  if (blocks > work_blocks) blocks = work_blocks;
  *grid = dim3((unsigned)(blocks < 1 ? 1 : blocks));
  return NTRU_OK;
}

// The dynamic-LDS limit of a kernel is raised at its first use, inside ntru_blocks_per_cu (resident_grid).
template <class Kern>
static int allow_lds(Kern, size_t) { return NTRU_OK; }

#define DISPATCH_K(Kv, ...)                                                                       \
  switch (Kv) {                                                                                     \
    case 1: { constexpr int KK = 1; __VA_ARGS__; } break;                                                  \
    case 3: { constexpr int KK = 3; __VA_ARGS__; } break;                                                  \
    case 5: { constexpr int KK = 5; __VA_ARGS__; } break;                                                  \
    case 7: { constexpr int KK = 7; __VA_ARGS__; } break;                                                  \
    case 9: { constexpr int KK = 9; __VA_ARGS__; } break;                                                  \
    case 11: { constexpr int KK = 11; __VA_ARGS__; } break;                                                \
    case 13: { constexpr int KK = 13; __VA_ARGS__; } break;                                                \
    case 15: { constexpr int KK = 15; __VA_ARGS__; } break;                                                \
    default: return fail(NTRU_ERR_UNSUPPORTED, "no kernel for this K");                             \
  }

static int check_common(const ntru_engine *eng, int N, int q, long B) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0) return fail(NTRU_ERR_ARG, "negative batch size");
  if (!is_pow2(q) || !ntru_engine_supports(N, q))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported (N, q): need 2 <= N <= 1920 and q a power of two <= 65536");
  return NTRU_OK;
}

static int check_pitch(int N, int ld) {
  if (ld < N || ld > 1024) return fail(NTRU_ERR_ARG, "row pitch must satisfy N <= ld <= 1024 elements");
  return NTRU_OK;
}
static const char *const kPitchedOnly = "a row pitch other than N needs the matrix-core kernels (kernel path 0 or 4, q <= 8192, N <= 1024, p == 3)";

extern "C" int ntru_encrypt_batch_dev(ntru_engine_t *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r,
                                      const uint8_t *d_m, int64_t B, uint16_t *d_e, uint16_t *d_quotE) {
  return ntru_encrypt_batch_pitched_dev(eng, N, q, N, d_h, d_r, d_m, B, d_e, d_quotE);
}

extern "C" int ntru_encrypt_batch_pitched_dev(ntru_engine_t *eng, int N, int q, int ld, const uint16_t *d_h,
                                              const uint8_t *d_r, const uint8_t *d_m, int64_t B, uint16_t *d_e,
                                              uint16_t *d_quotE) {
  if (int rc = check_common(eng, N, q, B)) return rc;
  if (ld != N) if (int rc = check_pitch(N, ld)) return rc;
  if (B == 0) return NTRU_OK;
  if (!d_h || !d_r || !d_m || !d_e) return fail(NTRU_ERR_ARG, "ntru_encrypt_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  Launch L;
  {
    MGeom mg;
    const size_t lds = make_mgeom(eng, N, q, ld, &mg)
                           ? (size_t)32 * mg.tpitch + (size_t)32 * mg.pitchA + (((size_t)32 * ld + 15) & ~(size_t)15) + 16 : 0;
    // role-split kernel (one workgroup of 8 waves per CU): needs 16-byte aligned batch arrays and its LDS to fit.  Only on
    // request for now: 1.9 ms per 2^20 at N = 821 against 1.5 ms for k_encrypt_m (profiles/r02_*role_split*).
    if (lds && eng->path == 5) {                           // lock-step variant: two four-wave groups per workgroup, one workgroup per CU
      const size_t per_group = (size_t)32 * mg.pitchA + (((size_t)32 * ld + 15) & ~(size_t)15) + 16;
      const size_t lds8 = (size_t)32 * mg.tpitch + 2 * per_group;
      if (lds8 <= 160 * 1024) {
        const long nrb = (long)((B + 31) / 32);
        if (int rc = resident_grid(eng, k_encrypt_m8, lds8, (nrb + 1) / 2, &L.grid, 2 * BLOCK_THREADS)) return rc;
        snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_m8");
        hipLaunchKernelGGL(k_encrypt_m8, L.grid, dim3(2 * BLOCK_THREADS), lds8, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, d_e, d_quotE);
        HIP_TRY(hipGetLastError());
        return NTRU_OK;
      }
    }
    if (lds && eng->path == 6 && ((((uintptr_t)d_r | (uintptr_t)d_m | (uintptr_t)d_e | (uintptr_t)d_quotE) & 15) == 0)) {
      size_t lds2 = (size_t)32 * mg.tpitch + (size_t)64 * mg.pitchA;
      for (int w = 0; w < 4; w++) lds2 += (size_t)32 * m2_chunk_pitch(mg.NT, w);
      if (lds2 <= 160 * 1024) {
        if (int rc = resident_grid(eng, k_encrypt_m2, lds2, (long)((B + 31) / 32), &L.grid, 512)) return rc;
        snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_m2");
        hipLaunchKernelGGL(k_encrypt_m2, L.grid, dim3(512), lds2, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, d_e, d_quotE);
        HIP_TRY(hipGetLastError());
        return NTRU_OK;
      }
    }
    if (lds && eng->path == 7 && 2 * (lds + WAVES_PER_BLOCK * OC_BYTES) <= 160 * 1024) {      // result chunks: aligned 16-byte stores
      const size_t ldsc = lds + WAVES_PER_BLOCK * OC_BYTES;
      if (int rc = resident_grid(eng, k_encrypt_mc, ldsc, (long)((B + 31) / 32), &L.grid)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_mc");
      hipLaunchKernelGGL(k_encrypt_mc, L.grid, dim3(BLOCK_THREADS), ldsc, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B,
                         d_e, d_quotE);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
    // The default: the operands reach LDS by direct-to-LDS loads, r of the next row block ahead of the last epilogue's stores:
    // 1.49-1.51 ms against 1.57-1.62 ms per 2^20 at N = 821 on the same device (profiles/r02_ab_direct_to_lds_rows.txt).
    // One direct-to-LDS instruction moves 64 x 16 bytes from the dword at or below a row, and eight of them per thread the m
    // image: a row of (its byte phase) + N > 1024 bytes, or an image of (phase) + 32 ld > 32768 bytes, would lose its last 1-3
    // bytes.  Those shapes (N >= 1022, or ld = 1024, with rows that are not dword-aligned) take k_encrypt_m, whose register
    // staging fetches the extra dword.
    const bool rows_dword_aligned = (ld & 3) == 0 && ((uintptr_t)d_r & 3) == 0, img_dword_aligned = (ld & 3) == 0 && ((uintptr_t)d_m & 3) == 0;
    const bool dma_fits = (N + 3 <= 1024 || (rows_dword_aligned && N <= 1024)) && (32 * ld + 3 <= 32768 || (img_dword_aligned && 32 * ld <= 32768));
    if (lds && (eng->path == 0 || eng->path == 8) && lds <= 160 * 1024 && dma_fits) {
      if (int rc = resident_grid(eng, k_encrypt_md, lds, (long)((B + 31) / 32), &L.grid)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_md");
      hipLaunchKernelGGL(k_encrypt_md, L.grid, dim3(BLOCK_THREADS), lds, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B,
                         d_e, d_quotE);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
    if (lds && lds <= 160 * 1024) {
      if (int rc = allow_lds(k_encrypt_m, lds)) return rc;
      if (int rc = resident_grid(eng, k_encrypt_m, lds, (long)((B + 31) / 32), &L.grid)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_m");
      hipLaunchKernelGGL(k_encrypt_m, L.grid, dim3(BLOCK_THREADS), lds, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B,
                         d_e, d_quotE);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
  }
  if (ld != N) return fail(NTRU_ERR_UNSUPPORTED, kPitchedOnly);
  if (const int me = add_path_me(eng, N, q)) {
    Geom g0 = make_geom(N, pick_K(N));
    if (int rc = plan_add(eng, N, B, (size_t)g0.eo_len * 8, (size_t)g0.nl * 4, &L)) return rc;
    DISPATCH_K_ADD(L.K, me, {
      if (int rc = allow_lds(k_encrypt_t<KK, MM>, L.lds)) return rc;
    if (int rc = resident_grid(eng, k_encrypt_t<KK, MM>, L.lds, (long)L.grid.x, &L.grid)) return rc;
      note_kernel(eng, "k_encrypt_t", KK, MM);
      hipLaunchKernelGGL((k_encrypt_t<KK, MM>), L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)q,
                         d_h, d_r, d_m, (long)B, d_e, d_quotE);
    });
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
  if (int rc = plan(eng, N, B, 1, false, &L)) return rc;
  DISPATCH_K(L.K, {
    if (int rc = allow_lds(k_encrypt<KK>, L.lds)) return rc;
    if (int rc = resident_grid(eng, k_encrypt<KK>, L.lds, (long)L.grid.x, &L.grid)) return rc;
    note_kernel(eng, "k_encrypt", KK, -1);
    hipLaunchKernelGGL(k_encrypt<KK>, L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)q, d_h, d_r, d_m,
                       (long)B, d_e, d_quotE);
  });
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

extern "C" int ntru_decrypt_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f,
                                      const uint8_t *d_fp, const uint16_t *d_e, int64_t B, uint8_t *d_value,
                                      uint16_t *d_quot1, uint16_t *d_rem1, uint8_t *d_quot2) {
  return ntru_decrypt_batch_pitched_dev(eng, N, q, p, N, d_f, d_fp, d_e, B, d_value, d_quot1, d_rem1, d_quot2);
}

extern "C" int ntru_decrypt_batch_pitched_dev(ntru_engine_t *eng, int N, int q, int p, int ld, const int8_t *d_f,
                                              const uint8_t *d_fp, const uint16_t *d_e, int64_t B, uint8_t *d_value,
                                              uint16_t *d_quot1, uint16_t *d_rem1, uint8_t *d_quot2) {
  if (int rc = check_common(eng, N, q, B)) return rc;
  if (ld != N) if (int rc = check_pitch(N, ld)) return rc;
  if (is_pow2(p) || !ntru_engine_supports(N, p))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported p: need a small non-power-of-two modulus with N*(p-1)^2 < 65536");
  if (B == 0) return NTRU_OK;
  if (!d_f || !d_fp || !d_e || !d_value) return fail(NTRU_ERR_ARG, "ntru_decrypt_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  Launch L;
  {
    MGeom mg;
    const size_t lds = (p == 3 && make_mgeom(eng, N, q, ld, &mg))
                           ? (size_t)32 * mg.tpitch + (size_t)64 * mg.pitchA + (size_t)256 * mg.NT + (((size_t)q + 15) & ~(size_t)15) : 0;
    // lock-step variant: one workgroup of two four-wave groups per CU, phases of the two groups interleaved by barriers.
    // The default where its LDS fits and a product takes two rounds of strips (N > 512): 2.52 against 2.63 ms per 2^20 at N = 821
    // (profiles/r02_ab_lockstep_phase_masks.txt), 2.16 against 2.24 ms at N = 701; at N = 509 (one round) it is 6 % slower, and so
    // it is without the witness arrays (shorter epilogues: 2.23 against 2.02 ms).
    if (lds && eng->path == 8) {                           // lock-step + direct-to-LDS loads of the next row block
      const size_t ldsd = (size_t)dec_dma_m3_bytes(N, p) + 2 * ((size_t)32 * dec_dma_row_pitch(mg.NT) + (size_t)256 * mg.NT) +
                          (size_t)32 * mg.tpitch + (((size_t)q + 15) & ~(size_t)15);
      if (ldsd <= 160 * 1024) {
        const long nrb = (long)((B + 31) / 32);
        if (int rc = resident_grid(eng, k_decrypt_m8d, ldsd, (nrb + 1) / 2, &L.grid, 2 * BLOCK_THREADS)) return rc;
        snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_m8d");
        hipLaunchKernelGGL(k_decrypt_m8d, L.grid, dim3(2 * BLOCK_THREADS), ldsd, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e,
                           (long)B, d_value, d_quot1, d_rem1, d_quot2);
        HIP_TRY(hipGetLastError());
        return NTRU_OK;
      }
    }
    if (lds && (eng->path == 5 || (eng->path == 0 && mg.NT > 16 && d_quot1 && d_rem1 && d_quot2))) {
      const size_t lds8 = 2 * ((size_t)64 * mg.pitchA + (size_t)256 * mg.NT) + (size_t)32 * mg.tpitch + (((size_t)q + 15) & ~(size_t)15);
      if (lds8 <= 160 * 1024) {
        const long nrb = (long)((B + 31) / 32);
        if (int rc = resident_grid(eng, k_decrypt_m8, lds8, (nrb + 1) / 2, &L.grid, 2 * BLOCK_THREADS)) return rc;
        snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_m8");
        hipLaunchKernelGGL(k_decrypt_m8, L.grid, dim3(2 * BLOCK_THREADS), lds8, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e,
                           (long)B, d_value, d_quot1, d_rem1, d_quot2);
        HIP_TRY(hipGetLastError());
        return NTRU_OK;
      }
    }
    if (lds && lds <= 160 * 1024) {
      if (int rc = allow_lds(k_decrypt_m, lds)) return rc;
      if (int rc = resident_grid(eng, k_decrypt_m, lds, (long)((B + 31) / 32), &L.grid)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_m");
      hipLaunchKernelGGL(k_decrypt_m, L.grid, dim3(BLOCK_THREADS), lds, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e,
                         (long)B, d_value, d_quot1, d_rem1, d_quot2);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
  }
  if (ld != N) return fail(NTRU_ERR_UNSUPPORTED, kPitchedOnly);
  {
    int me = 0;
    if (const int KS = shared_path_K(eng, N, q, p, &me)) {
      L.K = KS;
      L.g = make_geom(N, KS);
      const size_t per_wave = (size_t)L.g.G * (L.g.eo_len + 2) * 4;
      const int nblk8 = ((N + 31) >> 5) << 2;
      L.lds = (size_t)L.g.nl * 16 + (size_t)L.g.nl * KS * 8 + (size_t)nblk8 * 4 + WAVES_PER_BLOCK * per_wave;
      // product 2 on v_dot8 needs the 32-lane item layout (and is skipped when the MAC/add families are forced apart)
      const int d8 = (L.g.nl == 32 && eng->path != 3) ? 1 : 0;
      const long ngroups = (B + L.g.G - 1) / L.g.G;
      long blocks = (ngroups + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
      L.grid = dim3((unsigned)(blocks < 1 ? 1 : blocks));
      DISPATCH_K_SHARED(KS, me, d8, {
        if (int rc = allow_lds(k_decrypt_s<KK, MM, DD>, L.lds)) return rc;
        if (int rc = resident_grid(eng, k_decrypt_s<KK, MM, DD>, L.lds, (long)L.grid.x, &L.grid)) return rc;
        note_kernel(eng, DD ? "k_decrypt_s+dot8" : "k_decrypt_s", KK, MM);
        hipLaunchKernelGGL((k_decrypt_s<KK, MM, DD>), L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)q, (u32)p,
                           d_f, d_fp, d_e, (long)B, d_value, d_quot1, d_rem1, d_quot2);
      });
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
  }
  if (const int me = (p == 3 && eng->path == 2) ? add_path_me(eng, N, q) : 0) {   // per-item stepping: only when forced
    Geom g0 = make_geom(N, pick_K(N));
    if (int rc = plan_add(eng, N, B, (size_t)g0.eo_len * 8 + (size_t)g0.nl * 4,
                          (size_t)g0.eo_len * 8 + (size_t)g0.nl * 4, &L)) return rc;
    DISPATCH_K_ADD(L.K, me, {
      if (int rc = allow_lds(k_decrypt_t<KK, MM>, L.lds)) return rc;
    if (int rc = resident_grid(eng, k_decrypt_t<KK, MM>, L.lds, (long)L.grid.x, &L.grid)) return rc;
      note_kernel(eng, "k_decrypt_t", KK, MM);
      hipLaunchKernelGGL((k_decrypt_t<KK, MM>), L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)q, (u32)p,
                         d_f, d_fp, d_e, (long)B, d_value, d_quot1, d_rem1, d_quot2);
    });
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
  if (int rc = plan(eng, N, B, 2, false, &L)) return rc;
  DISPATCH_K(L.K, {
    if (int rc = allow_lds(k_decrypt<KK>, L.lds)) return rc;
    if (int rc = resident_grid(eng, k_decrypt<KK>, L.lds, (long)L.grid.x, &L.grid)) return rc;
    note_kernel(eng, "k_decrypt", KK, -1);
    hipLaunchKernelGGL(k_decrypt<KK>, L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)q, (u32)p, d_f, d_fp,
                       d_e, (long)B, d_value, d_quot1, d_rem1, d_quot2);
  });
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

extern "C" int ntru_polymul_split_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b,
                                      int64_t B, uint16_t *d_quot, uint16_t *d_rem) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0) return fail(NTRU_ERR_ARG, "negative batch size");
  if (!ntru_engine_supports(N, mod))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported (N, mod): mod must be a power of two <= 65536 or satisfy N*(mod-1)^2 < 65536");
  if (B == 0) return NTRU_OK;
  if (!d_a || !d_b || !d_quot || !d_rem) return fail(NTRU_ERR_ARG, "ntru_polymul_split: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  // per-item product on the matrix cores: mod a power of two <= 8192, 64 <= N <= 1024 (automatic from N = 128)
  if ((eng->path == 0 || eng->path >= 4) && is_pow2(mod) && mod <= 8192 && N <= 1024 && N >= (eng->path >= 4 ? 64 : 128)) {
    PGeom pg;
    pg.N = N; pg.NT = (N + 31) / 32; pg.tpitch = ((16 * pg.NT + 31) / 32) * 32 + 8;
    const size_t lds = PI_WAVES * pi_wave_bytes2(pg);
    if (int rc = allow_lds(k_polymul_m, lds)) return rc;
    int per_cu = 0;
    if (int rc = ntru_blocks_per_cu(eng, (const void *)k_polymul_m, 64 * PI_WAVES, lds, &per_cu)) return rc;
    long blocks = (long)eng->cus * (per_cu < 1 ? 1 : per_cu), work = (B + PI_WAVES - 1) / PI_WAVES;
    if (blocks > work) blocks = work;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_polymul_m");
    hipLaunchKernelGGL(k_polymul_m, dim3((unsigned)blocks), dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)mod, d_a, d_b,
                       (long)B, d_quot, d_rem);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
  Launch L;
  if (int rc = plan(eng, N, B, 0, true, &L)) return rc;
  DISPATCH_K(L.K, {
    if (int rc = allow_lds(k_polymul_split<KK>, L.lds)) return rc;
    if (int rc = resident_grid(eng, k_polymul_split<KK>, L.lds, (long)L.grid.x, &L.grid)) return rc;
    note_kernel(eng, "k_polymul_split", KK, -1);
    hipLaunchKernelGGL(k_polymul_split<KK>, L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)mod,
                       (int)is_pow2(mod), d_a, d_b, (long)B, d_quot, d_rem);
  });
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

// Per-item products with a ternary operand on the matrix cores: q a power of two <= 8192, 64 <= N <= 1024 (automatic
// from N = 128).
static bool product_tern_m_applies(const ntru_engine *eng, int N, int q) {
  return (eng->path == 0 || eng->path >= 4) && is_pow2(q) && q <= 8192 && N <= 1024 && N >= (eng->path >= 4 ? 64 : 128);
}
static int launch_product_tern_m(ntru_engine *eng, int N, int q, u32 mul, const u16 *d_a, const int8_t *d_s, long B,
                                 u16 *d_quot, u16 *d_rem) {
  PGeom pg;
  pg.N = N; pg.NT = (N + 31) / 32; pg.tpitch = ((16 * pg.NT + 31) / 32) * 32 + 8;
  const size_t lds = PI_WAVES * pi_wave_bytes(pg);
  if (int rc = allow_lds(k_product_tern_m, lds)) return rc;
  int per_cu = 0;
  if (int rc = ntru_blocks_per_cu(eng, (const void *)k_product_tern_m, 64 * PI_WAVES, lds, &per_cu)) return rc;
  long blocks = (long)eng->cus * (per_cu < 1 ? 1 : per_cu), work = (B + PI_WAVES - 1) / PI_WAVES;
  if (blocks > work) blocks = work;
  hipLaunchKernelGGL(k_product_tern_m, dim3((unsigned)blocks), dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)q, mul, d_a,
                     d_s, B, d_quot, d_rem);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

extern "C" int ntru_public_key_batch_dev(ntru_engine_t *eng, int N, int q, int p, const uint16_t *d_fq,
                                         const int8_t *d_g, int64_t B, uint16_t *d_h) {
  if (int rc = check_common(eng, N, q, B)) return rc;
  if (p < 1 || (long)p * (q - 1) >= 65536) return fail(NTRU_ERR_UNSUPPORTED, "p*(q-1) must fit 16 bits");
  if (B == 0) return NTRU_OK;
  if (!d_fq || !d_g || !d_h) return fail(NTRU_ERR_ARG, "ntru_public_key_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  if (product_tern_m_applies(eng, N, q)) {
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_public_key_m");
    return launch_product_tern_m(eng, N, q, (u32)p, d_fq, d_g, (long)B, nullptr, d_h);
  }
  Launch L;
  if (int rc = plan(eng, N, B, 0, true, &L)) return rc;
  DISPATCH_K(L.K, {
    if (int rc = allow_lds((k_polymul_split<KK, true>), L.lds)) return rc;
    if (int rc = resident_grid(eng, (k_polymul_split<KK, true>), L.lds, (long)L.grid.x, &L.grid)) return rc;
    note_kernel(eng, "k_public_key", KK, -1);
    hipLaunchKernelGGL((k_polymul_split<KK, true>), L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)q, 1,
                       (const u16 *)d_g, d_fq, (long)B, (u16 *)nullptr, d_h, (u32)p);
  });
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

extern "C" int ntru_verify_keys_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f,
                                          const int8_t *d_g, const uint16_t *d_fq, const uint8_t *d_fp,
                                          const uint16_t *d_h, int64_t B, uint16_t *d_quot_fq, uint16_t *d_rem_fq,
                                          uint8_t *d_quot_fp, uint8_t *d_rem_fp, uint16_t *d_quot_h,
                                          uint16_t *d_rem_h, uint8_t *d_flags) {
  if (int rc = check_common(eng, N, q, B)) return rc;
  if (is_pow2(p) || !ntru_engine_supports(N, p))
    return fail(NTRU_ERR_UNSUPPORTED, "unsupported p: need a small non-power-of-two modulus with N*(p-1)^2 < 65536");
  if ((long)(q - 1) * p > 65535) return fail(NTRU_ERR_UNSUPPORTED, "p*(q-1) must fit 16 bits");
  if (B == 0) return NTRU_OK;
  if (!d_f || !d_g || !d_fq || !d_fp || !d_h || !d_quot_fq || !d_rem_fq || !d_quot_fp || !d_rem_fp || !d_quot_h ||
      !d_rem_h || !d_flags)
    return fail(NTRU_ERR_ARG, "ntru_verify_keys_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  Launch L;
  // matrix-core kernel for per-item keys: p == 3, q <= 8192 (two int8 digit planes), 64 <= N <= 1024; automatic from N = 128
  if ((eng->path == 0 || eng->path >= 4) && p == 3 && q <= 8192 && N <= 1024 && N >= (eng->path >= 4 ? 64 : 128)) {
    PGeom pg;
    pg.N = N; pg.NT = (N + 31) / 32; pg.tpitch = ((16 * pg.NT + 31) / 32) * 32 + 8;
    const size_t lds = PI_WAVES * pi_wave_bytes(pg);
    if (int rc = allow_lds(k_verify_keys_m, lds)) return rc;
    int per_cu = 0;
    if (int rc = ntru_blocks_per_cu(eng, (const void *)k_verify_keys_m, 64 * PI_WAVES, lds, &per_cu)) return rc;
    long blocks = (long)eng->cus * (per_cu < 1 ? 1 : per_cu), work = (B + PI_WAVES - 1) / PI_WAVES;
    if (blocks > work) blocks = work;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_verify_keys_m");
    hipLaunchKernelGGL(k_verify_keys_m, dim3((unsigned)blocks), dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)q, d_f, d_g,
                       d_fq, d_fp, d_h, (long)B, d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp, d_quot_h, d_rem_h, d_flags);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
  if (const int me = p == 3 ? add_path_me(eng, N, q) : 0) {
    L.K = pick_K(N);
    L.g = make_geom(N, L.K);
    const size_t raw_len = ((size_t)N + 1) & ~(size_t)1;
    L.lds = WAVES_PER_BLOCK * ((size_t)L.g.eo_len * 8 + (size_t)L.g.nl * 4 + raw_len * 2);
    L.grid = dim3((unsigned)((B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK));
    DISPATCH_K_ADD(L.K, me, {
      if (int rc = allow_lds(k_verify_keys_t<KK, MM>, L.lds)) return rc;
      if (int rc = resident_grid(eng, k_verify_keys_t<KK, MM>, L.lds, (long)L.grid.x, &L.grid)) return rc;
      note_kernel(eng, "k_verify_keys_t", KK, MM);
      hipLaunchKernelGGL((k_verify_keys_t<KK, MM>), L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)q, (u32)p,
                         d_f, d_g, d_fq, d_fp, d_h, (long)B, d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp, d_quot_h, d_rem_h,
                         d_flags);
    });
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
  if (int rc = plan(eng, N, B, 0, true, &L)) return rc;
  DISPATCH_K(L.K, {
    if (int rc = allow_lds(k_verify_keys<KK>, L.lds)) return rc;
    if (int rc = resident_grid(eng, k_verify_keys<KK>, L.lds, (long)L.grid.x, &L.grid)) return rc;
    note_kernel(eng, "k_verify_keys", KK, -1);
    hipLaunchKernelGGL(k_verify_keys<KK>, L.grid, dim3(BLOCK_THREADS), L.lds, eng->stream, L.g, (u32)q, (u32)p, d_f, d_g,
                       d_fq, d_fp, d_h, (long)B, d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp, d_quot_h, d_rem_h, d_flags);
  });
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

static int check_elementwise(const ntru_engine *eng, int N, int mod, long B) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0) return fail(NTRU_ERR_ARG, "negative batch size");
  if (N < 1 || mod < 2 || mod > 65536) return fail(NTRU_ERR_UNSUPPORTED, "need N >= 1 and 2 <= mod <= 65536");
  return NTRU_OK;
}

static dim3 elementwise_grid(const ntru_engine *eng, long total) {
  long blocks = (total + 255) / 256, cap = (long)eng->cus * 8;
  return dim3((unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks)));
}

extern "C" int ntru_split_by_I_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, int64_t B,
                                   uint16_t *d_quot, uint16_t *d_rem) {
  if (int rc = check_elementwise(eng, N, mod, B)) return rc;
  if (B == 0) return NTRU_OK;
  if (!d_a || !d_quot || !d_rem) return fail(NTRU_ERR_ARG, "ntru_split_by_I: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  hipLaunchKernelGGL(k_split_by_I, elementwise_grid(eng, B * N), dim3(256), 0, eng->stream, N, (u32)mod, d_a, (long)B,
                     d_quot, d_rem);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

extern "C" int ntru_add_batch_dev(ntru_engine_t *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b,
                                  int64_t B, uint16_t *d_out) {
  if (int rc = check_elementwise(eng, N, mod, B)) return rc;
  if (B == 0) return NTRU_OK;
  if (!d_a || !d_b || !d_out) return fail(NTRU_ERR_ARG, "ntru_add_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  const long total = (long)B * N;
  const bool aligned = (((uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)d_out) & 15) == 0;
  const long nvec = aligned ? total / 8 : 0;
  if (nvec) {
    if (is_pow2(mod))
      hipLaunchKernelGGL(k_add_mod_vec<true>, elementwise_grid(eng, nvec), dim3(256), 0, eng->stream, (u32)mod,
                         (const u16x8 *)d_a, (const u16x8 *)d_b, nvec, (u16x8 *)d_out);
    else
      hipLaunchKernelGGL(k_add_mod_vec<false>, elementwise_grid(eng, nvec), dim3(256), 0, eng->stream, (u32)mod,
                         (const u16x8 *)d_a, (const u16x8 *)d_b, nvec, (u16x8 *)d_out);
  }
  if (nvec * 8 < total)
    hipLaunchKernelGGL(k_add_mod, elementwise_grid(eng, total - nvec * 8), dim3(256), 0, eng->stream, (u32)mod, d_a, d_b,
                       nvec * 8, total, d_out);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

// ---- key inversion, sampler, field packing: *_dev entry points (the host-pointer forms are in ntru_host.hip) --------

static const int64_t INVERT_CHUNK = 1 << 16;   // keys per set of Newton temporaries

template <int P, int NWC>
static int launch_invert_nw(ntru_engine *eng, int N, const int8_t *d_f, long B, uint16_t *d16, uint8_t *d8, uint8_t *d_flags,
                            unsigned bit) {
  const size_t lds = NWC ? 0 : (size_t)(P == 2 ? 4 : 8) * ((N + 32) / 32) * 64 * 4;
  if (lds > 160 * 1024) return fail(NTRU_ERR_UNSUPPORTED, "N too large for the inversion kernel's LDS planes");
  int per_cu = 0;
  if (int rc = ntru_blocks_per_cu(eng, (const void *)k_invert_key<P, NWC>, 64, lds, &per_cu)) return rc;
  long blocks = (B + 63) / 64, cap = (long)eng->cus * (per_cu < 1 ? 1 : per_cu);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((k_invert_key<P, NWC>), dim3((unsigned)blocks), dim3(64), lds, eng->stream, N, d_f, B, (u16 *)d16, d8,
                     d_flags, (u32)bit);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

// Register-resident planes for the word counts below (N + 1 bits rounded up to the next size), LDS planes otherwise.
template <int P>
static int launch_invert(ntru_engine *eng, int N, const int8_t *d_f, long B, uint16_t *d16, uint8_t *d8, uint8_t *d_flags,
                         unsigned bit) {
  const int nw = (N + 32) / 32;
#define INV_CASE(W) if (nw <= W) return launch_invert_nw<P, W>(eng, N, d_f, B, d16, d8, d_flags, bit);
  INV_CASE(2) INV_CASE(6) INV_CASE(12) INV_CASE(16) INV_CASE(22) INV_CASE(26)
  if constexpr (P == 2) { INV_CASE(32) }
#undef INV_CASE
  return launch_invert_nw<P, 0>(eng, N, d_f, B, d16, d8, d_flags, bit);
}

extern "C" int ntru_invert_key_batch_dev(ntru_engine_t *eng, int N, int q, int p, const int8_t *d_f, int64_t B,
                                         uint16_t *d_fq, uint8_t *d_fp, uint8_t *d_flags) {
  if (int rc = check_common(eng, N, q, B)) return rc;
  if (p != 3) return fail(NTRU_ERR_UNSUPPORTED, "key inversion implements p = 3 (and q a power of two)");
  if (B == 0) return NTRU_OK;
  if (!d_f || (!d_fq && !d_fp) || !d_flags) return fail(NTRU_ERR_ARG, "ntru_invert_key_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  HIP_TRY(hipMemsetAsync(d_flags, 0, (size_t)B, eng->stream));
  if (!d_fq) {                                     // only the inverse modulo p was asked for (polyInv(f, I, 3))
    if (int rc = launch_invert<3>(eng, N, d_f, (long)B, nullptr, d_fp, d_flags, NTRU_FLAG_NOT_UNIT_MODP)) return rc;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_invert_key");
    return NTRU_OK;
  }
  // mod 2 inverse straight into d_fq (as 0/1 coefficients), then Newton rounds v <- 2v - f v^2 mod q (index.js:499-506;
  // the reference runs log2(q) - 1 of them, the unique inverse mod q is reached once 2^rounds >= log2(q))
  if (int rc = launch_invert<2>(eng, N, d_f, (long)B, d_fq, nullptr, d_flags, NTRU_FLAG_NOT_UNIT_MOD2)) return rc;
  int k = 0;
  while ((1 << k) < q) k++;
  int rounds = 0;
  while ((1 << rounds) < k) rounds++;
  if (rounds > 0) {
    const int64_t C = B < INVERT_CHUNK ? B : INVERT_CHUNK;     // temporaries for C keys at a time
    const size_t row = (size_t)N * 2, part = ((size_t)C * row + 255) & ~(size_t)255;
    if (int rc = ntru_grow_dev(eng->cur_scratch, 4 * part)) return rc;      // engine-owned, grown on demand, never per call
    char *const sc = (char *)eng->cur_scratch->p;
    struct { void *p; } f16{sc}, t{sc + part}, u{sc + 2 * part}, qs{sc + 3 * part};
    for (int64_t o = 0; o < B; o += C) {
      const int64_t n = B - o < C ? B - o : C;
      uint16_t *v = d_fq + o * N;
      hipLaunchKernelGGL(k_signed_to_u16, elementwise_grid(eng, n * N), dim3(256), 0, eng->stream, d_f + o * N, (long)(n * N),
                         (u32)q, (u16 *)f16.p);
      for (int r = 0; r < rounds; r++) {
        // Hensel lifting: round r only has to be right modulo 2^(2^(r+1)); the early rounds therefore run modulo 4, 16,
        // 256 (single int8 digit planes on the matrix cores), the last one modulo q.  The inverse modulo q is unique.
        const int mr = (2 << r) >= k ? q : 1 << (2 << r);
        if (int rc = ntru_polymul_split_dev(eng, N, mr, v, v, n, (uint16_t *)qs.p, (uint16_t *)t.p)) return rc;
        if (product_tern_m_applies(eng, N, mr)) {         // f * t with f ternary: per-item product on the matrix cores
          if (int rc = launch_product_tern_m(eng, N, mr, 1u, (const u16 *)t.p, d_f + o * N, (long)n, nullptr, (u16 *)u.p)) return rc;
        } else if (int rc = ntru_polymul_split_dev(eng, N, mr, (const uint16_t *)f16.p, (const uint16_t *)t.p, n,
                                                   (uint16_t *)qs.p, (uint16_t *)u.p)) return rc;
        hipLaunchKernelGGL(k_newton_combine, elementwise_grid(eng, n * N), dim3(256), 0, eng->stream, (u16 *)v,
                           (const u16 *)u.p, (long)(n * N), (u32)mr);
      }
      HIP_TRY(hipGetLastError());                          // the next chunk reuses the temporaries in stream order
    }
  }
  if (d_fp) if (int rc = launch_invert<3>(eng, N, d_f, (long)B, nullptr, d_fp, d_flags, NTRU_FLAG_NOT_UNIT_MODP)) return rc;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_invert_key");
  return NTRU_OK;
}

static int sampler_pitch(int N) { int pd = (N + 15) / 16; if ((pd & 1) == 0) pd++; return pd; }   // dwords of 16 symbols

extern "C" int ntru_sample_ternary_dev(ntru_engine_t *eng, int N, int n1, int n2, int other, const uint32_t *key,
                                       uint64_t first_item, int64_t B, uint8_t *d_out) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0 || N < 1 || n1 < 0 || n2 < 0) return fail(NTRU_ERR_ARG, "negative size");
  if (n1 + n2 > N) return fail(NTRU_ERR_ARG, "The total of 1s and -1s cannot exceed the array length.");   // index.js:463
  if (other < 0 || other > 255) return fail(NTRU_ERR_ARG, "`other` must fit a byte");
  if (!key) return fail(NTRU_ERR_ARG, "ntru_sample_ternary: key is NULL");
  if (B == 0) return NTRU_OK;
  if (!d_out) return fail(NTRU_ERR_ARG, "ntru_sample_ternary: NULL buffer");
  const int pitch = sampler_pitch(N);
  const size_t lds = (size_t)64 * pitch * 4 + (size_t)((N + 2) & ~1) * 4;
  if (lds > 160 * 1024) return fail(NTRU_ERR_UNSUPPORTED, "N too large for the sampler's LDS rows");
  HIP_TRY(hipSetDevice(eng->device));
  ChaChaKey ck;
  memcpy(ck.k, key, 32);
  int per_cu = 0;
  if (int rc = ntru_blocks_per_cu(eng, (const void *)k_sample_ternary, 64, lds, &per_cu)) return rc;
  long blocks = (B + 63) / 64, cap = (long)eng->cus * (per_cu < 1 ? 1 : per_cu);
  if (blocks > cap) blocks = cap;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_sample_ternary");
  hipLaunchKernelGGL(k_sample_ternary, dim3((unsigned)blocks), dim3(64), lds, eng->stream, N, n1, n2, (u32)other, ck,
                     (unsigned long long)first_item, (long)B, d_out, pitch);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

extern "C" int ntru_pack_params(int max_val, int data_len, int *bits, int *per_output, int *arr_len, int *output_size) {
  if (max_val < 1 || max_val > 65535 || data_len < 0 || !bits || !per_output || !arr_len || !output_size)
    return fail(NTRU_ERR_ARG, "ntru_pack_params: need 1 <= max_val <= 65535, data_len >= 0 and non-NULL outputs");
  int b = 0;
  while ((max_val >> b) != 0) b++;                    // floor(log2(maxVal) + 1), index.js:573
  const int n = 252 / b;                              // index.js:574
  int al = ((data_len + n - 1) / n) * n;              // index.js:575-578
  if (al < 3 * n) al = 3 * n;
  int os = (al + n - 1) / n;                          // index.js:580
  if (os < 3) os = 3;
  *bits = b; *per_output = n; *arr_len = al; *output_size = os;
  return NTRU_OK;
}

extern "C" int ntru_pack_batch_dev(ntru_engine_t *eng, int max_val, int data_len, const uint16_t *d_data, int64_t B,
                                   uint64_t *d_out) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0) return fail(NTRU_ERR_ARG, "negative batch size");
  int bits, per, al, os;
  if (int rc = ntru_pack_params(max_val, data_len, &bits, &per, &al, &os)) return rc;
  if (B == 0) return NTRU_OK;
  if ((!d_data && data_len) || !d_out) return fail(NTRU_ERR_ARG, "ntru_pack_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  hipLaunchKernelGGL(k_pack, elementwise_grid(eng, B * os * 4), dim3(256), 0, eng->stream, bits, per, data_len, os, d_data,
                     (long)B, (unsigned long long *)d_out);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

extern "C" int ntru_unpack_batch_dev(ntru_engine_t *eng, int max_val, int packed_bits, const uint64_t *d_in,
                                     int packed_size, int64_t B, uint16_t *d_out) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0 || packed_size < 0) return fail(NTRU_ERR_ARG, "negative size");
  if (max_val < 1 || max_val > 65535) return fail(NTRU_ERR_ARG, "need 1 <= max_val <= 65535");
  int bits = 0;
  while ((max_val >> bits) != 0) bits++;
  const int per = packed_bits / bits;
  if (per < 1 || per * bits > 256) return fail(NTRU_ERR_ARG, "packed_bits does not hold a whole number of values within 256 bits");
  if (B == 0 || packed_size == 0) return NTRU_OK;
  if (!d_in || !d_out) return fail(NTRU_ERR_ARG, "ntru_unpack_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  hipLaunchKernelGGL(k_unpack, elementwise_grid(eng, B * packed_size * per), dim3(256), 0, eng->stream, bits, per,
                     packed_size, (const unsigned long long *)d_in, (long)B, d_out);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}
