// valu_families.hip -- MI355X (gfx950): the vector-ALU kernel families 1-3 of the NTRU polynomial-ring engine ("one ciphertext per
// wavefront", the decomposition BASELINE.json's north_star describes) and the host functions that launch them.
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
//     lane per two steps, lane stride K entries (K odd => conflict-free, profiles/archive/r01_microbench_valu_lds.txt).
//   * the "broadcast" operand a is read from LDS two coefficients at a time and applied with op_sel splats.
//   * T[k] = sum_i a[i] bc[k-i] is the cyclic product = remainder; the low half c[k] of the LINEAR product (needed
//     for the quotient) is T's value just before the lane's own block of i plus an in-block
//     triangle; high = T - low; quotient = -high.  (tools/lane_model.py is the executable spec of this indexing.)
//
// No CPU fallback exists in this file: every entry point needs a HIP device.
#include "kernels_common.h"

struct Geom {
  int N;       // ring size
  int nl;      // lanes per item = ceil(N / 2K)
  int G;       // items per wave = 64 / nl
  int off;     // K*nl: position of logical entry u = 0 inside an EO array
  int eo_len;  // 2*K*nl 8-byte entries per EO array
  int a_len;   // K*nl dwords (2K*nl u16) per staged a-operand
};

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
// multiply VALU op issues at 4 cycles per wave while a plain v_add_u32 issues at 2 (profiles/archive/r01_microbench_valu_lds.txt).
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
// plain assignments into v_cndmask, which is far slower on gfx950: profiles/archive/r01_microbench_exec_rate.txt).
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
// (profiles/archive/r01_microbench_step_rate.txt).  The per-item operand (e, then the lifted message) is the window; it is kept
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


// ---- host side: launchers of the vector-ALU families ---------------------------------------------------------------------------

static int pick_K(int N) {
  for (int K = 1; K <= 15; K += 2)
    if ((N + 2 * K - 1) / (2 * K) <= 64) return K;
  return 0;
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

int ntru_launch_encrypt_valu(ntru_engine *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m, int64_t B,
                             uint16_t *d_e, uint16_t *d_quotE) {
  Launch L;
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

int ntru_launch_decrypt_valu(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const uint8_t *d_fp, const uint16_t *d_e,
                             int64_t B, uint8_t *d_value, uint16_t *d_quot1, uint16_t *d_rem1, uint8_t *d_quot2) {
  Launch L;
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

int ntru_launch_polymul_valu(ntru_engine *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b, int64_t B, uint16_t *d_quot,
                             uint16_t *d_rem) {
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

int ntru_launch_public_key_valu(ntru_engine *eng, int N, int q, int p, const uint16_t *d_fq, const int8_t *d_g, int64_t B,
                                uint16_t *d_h) {
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

int ntru_launch_verify_keys_valu(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const int8_t *d_g, const uint16_t *d_fq,
                                 const uint8_t *d_fp, const uint16_t *d_h, int64_t B, uint16_t *d_quot_fq, uint16_t *d_rem_fq,
                                 uint8_t *d_quot_fp, uint8_t *d_rem_fp, uint16_t *d_quot_h, uint16_t *d_rem_h, uint8_t *d_flags) {
  Launch L;
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

