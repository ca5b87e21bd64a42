// matrix_common.h -- family 4: what the matrix-core translation units share (tile geometry, Toeplitz key arrays, the strip loop,
// aligned row loads, buffer descriptors).  Included by matrix_encrypt.hip, matrix_decrypt.hip and matrix_peritem.hip.
#ifndef NTRU_MATRIX_COMMON_H
#define NTRU_MATRIX_COMMON_H

#include "kernels_common.h"

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
// fragment per step).  tools/mfma_model.py is the executable specification; profiles/archive/r01_microbench_mfma_lds.txt holds
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
static __device__ unsigned long long g_stamps[1024][8][STAMP_BLOCKS][STAMP_SLOTS];     // [workgroup][wave: 8 in the lock-step kernels]; one copy per translation unit
#define STAMP(slot)                                                                                          \
  do {                                                                                                       \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024 && stamp_iter < STAMP_BLOCKS)                           \
      g_stamps[blockIdx.x][threadIdx.x >> 6][stamp_iter][slot] = __builtin_amdgcn_s_memtime();               \
  } while (0)
// every translation unit that stamps exports its own reader: NTRU_STAMPS_READER(ntru_debug_read_stamps_enc) ...
#define NTRU_STAMPS_READER(name) extern "C" int name(void *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
#else
#define STAMP(slot) do { } while (0)
#define NTRU_STAMPS_READER(name)
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
// diag(u): called once per sub-step u (0 .. NT_S-1; a constant once the block is unrolled) of the strip's DIAGONAL block, which every strip runs
// exactly once per product at a position that differs from wave to wave (contraction step kb0).  The row-image kernels
// (matrix_rowimage.hip) drain the previous row block's result images there: work with compile-time register indices, spread over
// the loop by the waves' different diagonals.
struct NoDiag { __device__ __forceinline__ void operator()(int) const {} };
// pause / pause_ib: pause() is called exactly once, before the first contraction step ib >= pause_ib is touched (at a
// block boundary, so possibly a few steps early; after the loops when no such step exists).  The role-split decrypt
// kernel waits there for the operand columns that are still being produced.
template <int MODE, int NT_S, class Epi, class Pause = NoPause, class Diag = NoDiag>
static __device__ __forceinline__ void toeplitz_strip(const unsigned char *__restrict__ st0,
                                                      const unsigned char *__restrict__ st1,
                                                      const u32 *__restrict__ tb0, const u32 *__restrict__ tb1,
                                                      const MGeom &g, int kb0, const u32 (&mlow)[4], Epi epi,
                                                      int stamp_iter = 0, int stamp_base = 0, int pause_ib = 0x7fffffff,
                                                      Pause pause = Pause(), Diag diag = Diag()) {
  constexpr bool TWO = MODE != M_DEC2;
  // The accumulators are never zeroed: the first matrix instruction of each takes the inline constant 0 as its C operand
  // (accL: contraction step 0, peeled below; accH: its own diagonal sub-step) -- 2 x 16 x NT_S moves per strip less.
  v16i accL[NT_S], accH[NT_S];
  v4i W0[NT_S], W1[NT_S];
  auto load_w = [&](int d, v4i &w0, v4i &w1) {
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
      // Tile 0 works on the fragment that was requested at the end of the sub-step before (slot (NT_S - u) % NT_S): it goes LAST, so
      // that the read has the other tiles' matrix instructions to land behind (as the first it exposed an LDS round trip per sub-step).
#pragma unroll
      for (int tt = 1; tt <= NT_S; tt++) {
        const int t = tt % NT_S;
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
      if (decltype(kind)::value == 2) diag(u);
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
  if (kb0 > 0 && g.NT > 0) {                              // contraction step 0: the first term of every `low`
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
  for (ib = kb0 + NT_S; ib + NT_S <= g.NT; ib += NT_S) { maybe_pause(ib, ib + NT_S - 1); block(ib, std::integral_constant<int, 1>{}); }   // below: high
  for (; ib < g.NT; ib++) { maybe_pause(ib, ib); single(ib, accH); }
  __builtin_amdgcn_s_setprio(0);
  if (!std::is_same<Pause, NoPause>::value && !paused) pause();
  STAMP(stamp_base);
  epi(accL, accH);
  STAMP(stamp_base + 1);
}

#ifdef NTRU_EXPERIMENTS
// ---- product of two SMALL-alphabet operands on the block-scaled fp4 matrix instruction -------------------------------------------------
// decryptBits' second product is ternary x ternary: both operands are exact in fp4 (e2m1: 0, 1, 2 = codes 0, 2, 4), the sums stay
// below 4 N < 2^24 (exact in the fp32 accumulators), and v_mfma_scale_f32_32x32x64_f8f6f4 contracts K = 64 per instruction in the
// clocks the int8 instruction takes for K = 32: half the matrix instructions, half the operand reads (scales 2^0).
// Same tile geometry (rows = the 32 items of a row block, 32-column tiles); a contraction BLOCK is 64 coefficients = two tiles:
// lane (r = lane & 31, h = lane >> 5) holds the 32 nibbles k = 32 h .. 32 h + 31 of its row (A) / column (B), low nibble first
// (both operands use the same order, which is all the instruction needs).  Operand stage: [32 rows][pitch4] bytes, two columns per
// byte.  Key array: reversed and cyclic as before, as a NIBBLE stream in 8 nibble-shifted copies (a lane's 32-nibble fragment
// starts at an arbitrary nibble; copy y & 7 holds it as 4 aligned dwords): rev4[y] = code(s[(32 NT - 1 - y) mod N]).
// Tile kb against block s: G = kb - 2 s >= 2 -> all of it is `low`, G < 0 -> `high`, G = 0 / 1 -> the diagonal runs through the
// lower / upper half of the block (two instructions with complementary nibble masks).
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
struct FGeom { int NT2, pitch4, tp4; };
static __host__ __device__ inline FGeom make_fgeom(int NT) {
  FGeom f;
  f.NT2 = (NT + 1) / 2;
  f.pitch4 = 32 * f.NT2 + 16;                              // an odd multiple of 16 bytes: conflict-free ds_read_b128
  const int L = 32 * NT + 64 * f.NT2 + 64;                 // nibbles a fragment request can touch (one block of run-ahead included)
  f.tp4 = ((L / 8 + 4 + 31) / 32) * 32 + 4;                // dwords per copy
  return f;
}
static __host__ __device__ inline size_t fp4_array_bytes(int NT) { return (size_t)32 * make_fgeom(NT).tp4; }

template <class D>
static __device__ __forceinline__ void build_toeplitz_array4(u32 *T4, const MGeom &g, const FGeom &f4, D value, int tid, int nthr) {
  const int Y0 = 32 * g.NT - 1;
  for (int x = tid; x < 8 * f4.tp4; x += nthr) {
    const int c = x / f4.tp4, w = x - c * f4.tp4;
    u32 v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int idx = (Y0 - (8 * w + c + j)) % g.N;             // remainder takes the dividend's sign
      idx += idx < 0 ? g.N : 0;
      v |= (((u32)value(idx) & 3u) << 1) << (4 * j);      // fp4 e2m1 codes of 0, 1, 2
    }
    T4[x] = v;
  }
}
// Per-lane pointer to the fragment of G = 0 (the fragment of G starts 4 G dwords below it).
static __device__ __forceinline__ const u32 *frag4_lane_base(const u32 *T4, const MGeom &g, const FGeom &f4, int lane) {
  const int yb = 32 * g.NT - 1 - (lane & 31) + 32 * (lane >> 5);
  return T4 + (yb & 7) * f4.tp4 + (yb >> 3);
}
// Nibble masks of the `low` part when the diagonal runs through this block: m0: G = 0 (lower half diagonal, upper half all high),
// m1: G = 1 (lower half all low, upper half diagonal).  Nibble j of a lane is contraction index j of its half; low iff j <= column.
static __device__ __forceinline__ void diag_low_mask4(int lane, u32 (&m0)[4], u32 (&m1)[4]) {
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    u32 mk = 0;
#pragma unroll
    for (int jj = 0; jj < 8; jj++) mk |= (8 * c + jj <= n) ? (0xFu << (4 * jj)) : 0u;
    m0[c] = h == 0 ? mk : 0u;
    m1[c] = h == 0 ? 0xFFFFFFFFu : mk;
  }
}

// One strip of NT_S <= 4 column tiles from tile kb0, every contraction block; st: this lane's row of the nibble stage (+ 16 bytes for
// the upper half-wave), tb4: its fragment base.  epi(lo, hi) gets the two halves as int32 tiles, like toeplitz_strip's.
// Three regions, each with compile-time kinds per tile (runtime kinds cost a second copy of every accumulator): blocks below the
// strip's diagonals (all `low`, a loop), the 1-3 blocks the diagonals run through (unrolled; two variants by the parity of kb0),
// blocks above (all `high`, a loop).  Fragment window: inside a region tile t works on slot (t + 2 parity) & 3 and the fragments
// tiles 0, 1 need in the NEXT block go into the slots tiles 2, 3 have just used (G drops by two per block, so tiles 2, 3 inherit the
// fragments of tiles 0, 1); a region of odd length ends with the two slot pairs exchanged, so every region starts at parity 0.
template <int NT_S, class Epi>
static __device__ __forceinline__ void toeplitz_strip_fp4(const unsigned char *__restrict__ st, const u32 *__restrict__ tb4,
                                                          const FGeom &f4, int kb0, int lane_, Epi epi) {
  v16f accL[NT_S], accH[NT_S];
#pragma unroll
  for (int t = 0; t < NT_S; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) { accL[t][i] = 0.f; accH[t][i] = 0.f; }
  v4i W[4];
  const int Gmin = -2 * f4.NT2;
  auto load_w = [&](int G, v4i &w) {
    G = G < Gmin ? Gmin : G;                               // run-ahead past the last block: inside the array, never used
    const u32 *p = tb4 - 4 * G;
    w = (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
  };
  auto mm = [&](v16f &acc, const v4i &a, const v4i &w) {
    const v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, w8 = {w[0], w[1], w[2], w[3], 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8, acc, 4, 4, 0, 127, 0, 127);   // fp4 x fp4, scales 2^0
  };
#pragma unroll
  for (int t = 0; t < NT_S; t++) load_w(kb0 + t, W[t]);
  for (int t = NT_S; t < 4; t++) W[t] = (v4i){0, 0, 0, 0};
  v4i a = *(const v4i *)st;
  enum { LOW = 0, HIGH = 1, MIX0 = 2, MIX1 = 3 };
  u32 m0[4], m1[4];                                        // nibble masks of the diagonal blocks: made right before those blocks
  auto tile = [&](auto kind, int t, int slot, const v4i &av) {
    constexpr int K = decltype(kind)::value;
    if (K == LOW) mm(accL[t], av, W[slot]);
    else if (K == HIGH) mm(accH[t], av, W[slot]);
    else {
      v4i wl, wh;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const u32 mk = K == MIX0 ? m0[c] : m1[c];
        wl[c] = (int)((u32)W[slot][c] & mk); wh[c] = (int)((u32)W[slot][c] & ~mk);
      }
      mm(accL[t], av, wl);
      mm(accH[t], av, wh);
    }
  };
#define FP4_KIND(G) std::integral_constant<int, ((G) >= 2 ? LOW : ((G) < 0 ? HIGH : ((G) == 0 ? MIX0 : MIX1)))>{}
  // one block; FIX = 0: every tile low, 1: every tile high, 2: tile t has G = G0 + t (G0 a compile-time constant)
  auto block = [&](int s, auto par, auto fix, auto g0) {
    constexpr int P = decltype(par)::value, FIX = decltype(fix)::value, G0 = decltype(g0)::value;
    const v4i an = *(const v4i *)(st + 32 * (s + 1));      // (past the last block: the row's pad / the next row, never used)
    if (NT_S > 2) tile(FP4_KIND(FIX == 0 ? 2 : (FIX == 1 ? -1 : G0 + 2)), 2, (2 + 2 * P) & 3, a);
    if (NT_S > 3) tile(FP4_KIND(FIX == 0 ? 2 : (FIX == 1 ? -1 : G0 + 3)), 3, (3 + 2 * P) & 3, a);
    load_w(kb0 - 2 * (s + 1), W[(2 * P + 2) & 3]);
    if (NT_S > 1) load_w(kb0 + 1 - 2 * (s + 1), W[(2 * P + 3) & 3]);
    tile(FP4_KIND(FIX == 0 ? 2 : (FIX == 1 ? -1 : G0 + 0)), 0, (0 + 2 * P) & 3, a);
    if (NT_S > 1) tile(FP4_KIND(FIX == 0 ? 2 : (FIX == 1 ? -1 : G0 + 1)), 1, (1 + 2 * P) & 3, a);
    a = an;
  };
  auto swap_pairs = [&]() { const v4i x0 = W[0], x1 = W[1]; W[0] = W[2]; W[1] = W[3]; W[2] = x0; W[3] = x1; };
  using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
  auto uniform_region = [&](int s0, int s1, auto fix) {    // blocks s0 .. s1-1, all tiles of one kind
    int s = s0;
    for (; s + 1 < s1; s += 2) { block(s, C0{}, fix, C0{}); block(s + 1, C1{}, fix, C0{}); }
    if (s < s1) { block(s, C0{}, fix, C0{}); swap_pairs(); }
  };
  __builtin_amdgcn_s_setprio(3);
  const int sa = kb0 >> 1;
  uniform_region(0, sa, C0{});
  int s_next;
  asm volatile("" : "+v"(lane_));                          // (made here: eight registers that need not live through the loops)
  diag_low_mask4(lane_, m0, m1);
  if ((kb0 & 1) == 0) {                                    // tile 0 has G = 0 in block sa
    block(sa, C0{}, C2{}, std::integral_constant<int, 0>{});
    if (NT_S > 2) { block(sa + 1, C1{}, C2{}, std::integral_constant<int, -2>{}); s_next = sa + 2; }
    else { swap_pairs(); s_next = sa + 1; }
  } else {                                                 // tile 0 has G = 1 in block sa
    block(sa, C0{}, C2{}, std::integral_constant<int, 1>{});
    if (NT_S > 1) {
      block(sa + 1, C1{}, C2{}, std::integral_constant<int, -1>{});
      if (NT_S > 3) { block(sa + 2, C0{}, C2{}, std::integral_constant<int, -3>{}); swap_pairs(); s_next = sa + 3; }
      else s_next = sa + 2;
    } else { swap_pairs(); s_next = sa + 1; }
  }
  uniform_region(s_next, f4.NT2, C1{});
  __builtin_amdgcn_s_setprio(0);
#undef FP4_KIND
  v16i lo[NT_S], hi[NT_S];
#pragma unroll
  for (int t = 0; t < NT_S; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) { lo[t][i] = (int)accL[t][i]; hi[t][i] = (int)accH[t][i]; }
  epi(lo, hi);
}

#endif   // NTRU_EXPERIMENTS (fp4 second product: bit-exact, measured 5 % slower, profiles/r04_fp4_product2.txt)

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
// address runs at a fraction of the aligned rate (profiles/archive/r01_ablation_mfma.txt: 0.8 of 2.5 ms), so rows are read as
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

// ---- host side ----------------------------------------------------------------------------------------------------------
// Matrix-core path (family 4): shared key, q a power of two <= 8192 (two int8 digit planes), LDS for a 32-row block.
static inline bool make_mgeom(const ntru_engine *eng, int N, int q, int ld, MGeom *g) {
  if (eng->path != 0 && eng->path < 4) return false;
  if (q > 8192 || N > 1024 || ld > 1024 || N < (eng->path >= 4 || ld != N ? 2 : 64)) return false;   // staging: lane = 16-byte chunk of a row
  g->N = N;
  g->ld = ld;
  g->NT = (N + 31) / 32;
  g->pitchA = 32 * g->NT + 16;
  g->tpitch = ((16 * g->NT + 31) / 32) * 32 + 8;
  return true;
}


static const char *const kMatrixLdsLimitNote = "160 KB of LDS per CU";

#endif
