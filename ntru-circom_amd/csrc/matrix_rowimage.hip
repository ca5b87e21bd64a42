// matrix_rowimage.hip -- MI355X (gfx950), family 4, "row image" kernels: the same batch x Toeplitz products as matrix_encrypt.hip /
// matrix_decrypt.hip (v_mfma_i32_32x32x32_i8, exact), with a different way OUT of the CU.
//
// Why.  The result arrays are dense [B][N] rows with N odd, so no row starts on a 16-byte boundary and a tile of the accumulator
// layout (lane = column, register = row) leaves as 2-byte stores, two 64-byte pieces per instruction: 2.4-2.7 TB/s and 260 pJ per
// byte (profiles/archive/r03_power_store_patterns.txt, pattern 0) -- 46 % of decrypt's and 60 % of encrypt's dynamic energy at the 1400 W
// cap.  But the 32 rows of a row block are ONE contiguous run of 32 * 2N bytes in every result array, and a contiguous run written
// as aligned 16-byte pieces is the fastest and cheapest pattern there is (pattern 4: 4.9 TB/s, 135 pJ per byte).  These kernels
// build an exact IMAGE of that run in LDS -- byte i of the image is byte i of the row block's slice of the array, shifted so that
// 16-byte boundaries of LDS are 16-byte boundaries of global memory -- and copy it out flat: one ds_read_b128 + one
// buffer_store_dwordx4 per 16 bytes, the (at most two) partial pieces at the ends of the run as 2-byte stores.
//
// How it fits.  The image of a row block (one dword per element: low + high | high << 16, both result arrays come out of it) is
// 105 KB at N = 821; with the operand stage and the key arrays that leaves room for ONE row block per CU.  So: one workgroup of
// EIGHT waves per CU, every wave one strip of <= 4 column tiles (all 26 tiles in one round), and the copy-out of row block k runs
// INSIDE the matrix loops of row block k + 1: each wave drains its share of the pieces in the sub-steps of its strip's diagonal
// block (compile-time register indices; the eight diagonals sit at eight different contraction steps, so the stores are spread
// over the loop).  The epilogue shrinks to one add, one v_perm and one ds_write_b32 per accumulator element; "mod q", the negation
// and the plaintext add are packed 16-bit operations on the 8 elements of a piece at drain time (the plaintext bytes of a piece
// are 8 consecutive bytes of m, requested a whole loop ahead).
// Per row block: barrier (image + stage complete) -> loops [drain the previous image; next r rows requested into registers]
// -> barrier (stage and image free) -> epilogue into the image, next r rows into the stage.
// Dense rows only (ld == N), 2-byte aligned result arrays that share their 16-byte phase; everything else takes matrix_encrypt.hip.
//
// MEASURED (profiles/r04_rowimage_encrypt.txt): bit-exact, and exactly as fast as k_encrypt_md (1.50 against 1.51 ms per 2^20 at
// N = 821; 1.58 against 1.58 ms in a sustained loop) at exactly the same energy (1.74 J dynamic per launch, both at the 1400 W cap):
// inside a kernel the store pattern does not change what an HBM byte costs.  Kept as an experiment (kernel path 10 of a library built
// with -DNTRU_EXPERIMENTS).
#include "matrix_common.h"

typedef int v2i __attribute__((ext_vector_type(2)));

constexpr int RI_WAVES = 8, RI_THREADS = RI_WAVES * 64;

// Raw image of a row block: one dword per element, (low + high) & 0xffff | high << 16, element t = row * N + col at byte 2 a0 + 4 t,
// a0 = the 16-byte phase of the row block's slice in the uint16 result arrays (the same for every array that is written): the 8
// dwords behind output piece k (global bytes G0 + 16 k, elements 8 k - a0 / 2 ...) then are the image bytes [32 k, 32 k + 32).
static __host__ __device__ inline int ri_img_bytes(int N) { return (128 * N + 32 + 15) & ~15; }

static __device__ __forceinline__ void ri_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// What a drain needs to know about the row block whose image is in LDS.
struct RiPrev {
  long b0;          // first row
  int L;            // bytes of its slice of a uint16 array: 2 * rows * N
};

#ifdef NTRU_EXPERIMENTS
constexpr int RI_MAXPI = 8;                      // piece instructions (64 output pieces of 16 bytes) per wave: row blocks up to 64 KB per array
constexpr int RI_EDGE_WAVE = 7;                  // the wave that also stores the two partial pieces at the ends of a row block's run
struct RiRaw { v4i x0, x1; };
// The 8 raw dwords behind output piece k = 64 pi + lane (lanes past the end of the run read piece 0; they store nothing).
static __device__ __forceinline__ RiRaw ri_read_piece(const unsigned char *img, int pi, int lane, int end) {
  const int k = 64 * pi + lane;
  const unsigned char *p = img + (16 * k < end ? 32 * k : 0);
  RiRaw r;
  r.x0 = *(const v4i *)p;
  r.x1 = *(const v4i *)(p + 16);
  return r;
}
// Piece k of both result arrays of encryptBits: e = (raw.low + m) mod q, quotientE = (-raw.high) mod q, 8 elements each, as one
// aligned 16-byte store per array; partial pieces (the two ends of the run) are left to ri_encrypt_edges.
template <bool WQ>
static __device__ __forceinline__ void ri_store_encrypt(const RiRaw &rw, int pi, int lane, int a0, int end, u32 q, v2i mv,
                                                        const __amdgpu_buffer_rsrc_t &rsE, const __amdgpu_buffer_rsrc_t &rsQ) {
  const int pos = 16 * (64 * pi + lane);
  const int voff = pos >= a0 && pos + 16 <= end ? pos : (int)0x80000000;
  const u32 qm2 = (q - 1) * 0x00010001u;
  const u32 mm[4] = {__builtin_amdgcn_perm(0u, (u32)mv[0], 0x0c010c00u), __builtin_amdgcn_perm(0u, (u32)mv[0], 0x0c030c02u),
                     __builtin_amdgcn_perm(0u, (u32)mv[1], 0x0c010c00u), __builtin_amdgcn_perm(0u, (u32)mv[1], 0x0c030c02u)};
  const u32 x[8] = {(u32)rw.x0[0], (u32)rw.x0[1], (u32)rw.x0[2], (u32)rw.x0[3], (u32)rw.x1[0], (u32)rw.x1[1], (u32)rw.x1[2], (u32)rw.x1[3]};
  v4i ev, qv;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const u32 lowp = __builtin_amdgcn_perm(x[2 * c + 1], x[2 * c], 0x05040100u), highp = __builtin_amdgcn_perm(x[2 * c + 1], x[2 * c], 0x07060302u);
    ev[c] = (int)(as_u32(as_pair(lowp) + as_pair(mm[c])) & qm2);
    qv[c] = (int)(as_u32((u16x2){0, 0} - as_pair(highp)) & qm2);
  }
  __builtin_amdgcn_raw_buffer_store_b128(ev, rsE, voff, 0, ST_AUX);
  if (WQ) __builtin_amdgcn_raw_buffer_store_b128(qv, rsQ, voff, 0, ST_AUX);
}

// encryptBits (index.js:87-110) for a batch under one shared key; see the file header.  r in {0..3} bytes, h < q <= 8192 as in
// k_encrypt_m: planes [r | 32 r] x [d0 ; 4 d1].
__global__ __launch_bounds__(RI_THREADS, 2) void k_encrypt_w(MGeom g, u32 q, const u16 *__restrict__ h, const uint8_t *__restrict__ r,
                                                             const uint8_t *__restrict__ m, long B, u16 *__restrict__ e,
                                                             u16 *__restrict__ quotE) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u32 *T0 = (u32 *)lds, *T1 = T0 + 4 * g.tpitch;
  unsigned char *stA = (unsigned char *)(T1 + 4 * g.tpitch);
  unsigned char *img = stA + 32 * g.pitchA;
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int hthr = (int)(q >> 1) - 65;
  auto hs_of = [&](int i) { int hv = (int)(h[i] & (q - 1)); return hv > hthr ? hv - (int)q : hv; };
  build_toeplitz_array(T0, g, [&](int i) { const int hs = hs_of(i); return ((hs + 64) & 127) - 64; }, tid0, RI_THREADS);
  build_toeplitz_array(T1, g, [&](int i) { const int hs = hs_of(i); const int d0 = ((hs + 64) & 127) - 64; return ((hs - d0) >> 7) * 4; }, tid0, RI_THREADS);
  const bool want_q = quotE != nullptr;
  const long nrb = (B + 31) >> 5;
  // this wave's strip: the NT tiles cut into 8 strips as even as possible, in column order (waves w and w + 4 share a SIMD: the
  // wider strips go to waves 0 .. rem-1, i.e. to different SIMDs first)
  const int sbase = g.NT / RI_WAVES, srem = g.NT % RI_WAVES;
  const int nt = sbase + (wave < srem ? 1 : 0), kb0 = wave * sbase + (wave < srem ? wave : srem);
  // 16-byte phase of a row block's slice in the result arrays (the host has checked that e and quotE share it) and 4-byte phase of
  // its plaintext bytes: the same for every row block (a row block is 64 N bytes of a result array and 32 N bytes of m)
  const int a0 = __builtin_amdgcn_readfirstlane((int)((unsigned long long)e & 15));
  const int c0 = __builtin_amdgcn_readfirstlane((int)((unsigned long long)m & 3));
  constexpr int RPW = 32 / RI_WAVES;
  RawChunks<1> in_r[RPW];
  auto request_r = [&](long rb, int lane) {                  // rows wave, wave + 8, ...: lane = 16-byte chunk of a row (N <= 1024)
    const long b0 = rb << 5 < B ? rb << 5 : B;
    const AlignedSrc src_r = aligned_src(r + b0 * g.N, (B - b0) * g.N);
#pragma unroll
    for (int j = 0; j < RPW; j++) in_r[j] = load_raw<1>(src_r, src_r.a0 + (wave + RI_WAVES * j) * g.N + 16 * lane, 0);
  };
  auto stage_r = [&](long rb, int lane) {
    const long b0 = rb << 5 < B ? rb << 5 : B;
    const int ar = (int)((unsigned long long)(r + b0 * g.N) & 15);
    const v4i mk = col_mask16(16 * lane, g.N);
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int row = wave + RI_WAVES * j;
      v4i v[1];
      shift_raw<1>(in_r[j], ar + row * g.N, v);
      if (lane < 2 * g.NT) *(v4i *)(stA + row * g.pitchA + 16 * lane) = v[0] & mk;
    }
  };
  // Edge elements of a run (wave RI_EDGE_WAVE): lanes 0-7 = the elements of the head piece, lanes 8-15 = of the tail piece.
  auto edge_pos = [&](int lane, int end, bool *ok) {
    const int c = lane & 7, pos = (lane < 8 ? 0 : end & ~15) + 2 * c;
    *ok = lane < 16 && pos >= a0 && pos < end && (lane < 8 ? a0 != 0 : (end & 15) != 0);   // (a run shorter than a piece: stored twice, same values)
    return pos;
  };
  v2i mreg[RI_MAXPI];                                        // plaintext bytes of this wave's pieces
  u32 medge = 0;                                             // ... and of its edge element
  auto request_m = [&](const RiPrev &pv, int lane) {
    const int end = a0 + pv.L;
    const __amdgpu_buffer_rsrc_t rsm = rows_rsrc((const void *)((unsigned long long)(m + pv.b0 * g.N) & ~3ULL), c0 + (pv.L >> 1));
#pragma unroll
    for (int j = 0; j < RI_MAXPI; j++) {
      const int pos = 16 * (64 * (RI_WAVES * j + wave) + lane);
      const bool full = pos >= a0 && pos + 16 <= end;
      mreg[j] = (v2i){0, 0};
      if (1024 * (RI_WAVES * j + wave) < end)                // wave-uniform
        mreg[j] = __builtin_amdgcn_raw_buffer_load_b64(rsm, full ? c0 + ((pos - a0) >> 1) : (int)0x80000000, 0, 0);
    }
    if (wave == RI_EDGE_WAVE) {
      bool ok;
      const int pos = edge_pos(lane, end, &ok);
      medge = (unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsm, ok ? c0 + ((pos - a0) >> 1) : (int)0x80000000, 0, 0);
    }
  };
  RiPrev prev = {0, 0};
  bool have_prev = false;
  // Pieces 8 j + wave, j = j0 .. j1-1, of the previous row block: every LDS read of the group first, then the arithmetic and the stores.
  auto drain = [&](int j0, int j1, int lane) {
    RiPrev pv = prev;
    asm volatile("" : "+s"(pv.b0), "+s"(pv.L));
    const int end = a0 + pv.L;
    RiRaw rw[RI_MAXPI];
#pragma unroll
    for (int j = 0; j < RI_MAXPI; j++)
      if (j >= j0 && j < j1 && 1024 * (RI_WAVES * j + wave) < end) rw[j] = ri_read_piece(img, RI_WAVES * j + wave, lane, end);
    // the row block's slice as a buffer: base G0 = 16-byte boundary at or below its first element, a0 + L bytes (nothing below a0
    // is touched: those bytes belong to the row block before)
    const __amdgpu_buffer_rsrc_t rsE = rows_rsrc((const unsigned char *)(e + pv.b0 * g.N) - a0, end);
    const __amdgpu_buffer_rsrc_t rsQ = rows_rsrc((const unsigned char *)((want_q ? quotE : e) + pv.b0 * g.N) - a0, end);
#pragma unroll
    for (int j = 0; j < RI_MAXPI; j++)
      if (j >= j0 && j < j1 && 1024 * (RI_WAVES * j + wave) < end) {
        if (want_q) ri_store_encrypt<true>(rw[j], RI_WAVES * j + wave, lane, a0, end, q, mreg[j], rsE, rsQ);
        else ri_store_encrypt<false>(rw[j], RI_WAVES * j + wave, lane, a0, end, q, mreg[j], rsE, rsQ);
      }
    if (j1 == RI_MAXPI && wave == RI_EDGE_WAVE) {            // the (at most two) partial pieces of the run, element by element
      bool ok;
      const int pos = edge_pos(lane, end, &ok);
      const u32 x = *(const u32 *)(img + (ok ? 2 * pos : 0));   // element (pos - a0) / 2 lives at 2 a0 + 4 (pos - a0) / 2 = 2 pos
      __builtin_amdgcn_raw_buffer_store_b16((u16)((x + medge) & (q - 1)), rsE, ok ? pos : (int)0x80000000, 0, ST_AUX);
      if (want_q) __builtin_amdgcn_raw_buffer_store_b16((u16)((0u - (x >> 16)) & (q - 1)), rsQ, ok ? pos : (int)0x80000000, 0, ST_AUX);
    }
  };

  long rb = blockIdx.x;
  int stamp_iter = -1;
  (void)stamp_iter;
  if (rb < nrb) { request_r(rb, lane0); stage_r(rb, lane0); }
  for (; rb < nrb; rb += gridDim.x) {
    int lane = lane0, N = g.N;
    asm volatile("" : "+v"(lane), "+s"(N));
    stamp_iter++;
    STAMP(0);
    ri_barrier();                                        // B: stage of rb and image of `prev` complete (first pass: key arrays)
    STAMP(1);
    const long rb_next = rb + gridDim.x;
    if (rb_next < nrb) request_r(rb_next, lane);
    const u32 *tb0 = frag_lane_base(T0, g, lane), *tb1 = frag_lane_base(T1, g, lane);
    const unsigned char *st0 = stA + (lane & 31) * g.pitchA + 16 * (lane >> 5);
    u32 mlow[4];
    diag_low_mask(lane, mlow);
    const long b0 = rb << 5;
    const RiPrev cur = {b0, (int)(2 * ((B - b0 < 32 ? B - b0 : 32)) * N)};
    auto finish = [&](int lane_) {                           // after the loops
      STAMP(4);
      request_m(cur, lane_);                               // the plaintext bytes the NEXT loops will add while draining this row block
      STAMP(5);
      ri_barrier();                                      // A: every wave has left its loops -- stage free, image drained
      STAMP(6);
    };
    auto image = [&](auto &lo, auto &hi) {
      constexpr int NTS = sizeof(lo) / sizeof(lo[0]);
      const int colb = 32 * kb0 + (lane & 31), rowb = 4 * (lane >> 5);
#pragma unroll
      for (int t = 0; t < NTS; t++) {
        if (colb + 32 * t < N) {
          unsigned char *pe = img + 2 * a0 + 4 * (rowb * N + colb + 32 * t);
#pragma unroll
          for (int i = 0; i < 16; i++)
            *(u32 *)(pe + 4 * ((i & 3) + 8 * (i >> 2)) * N) = __builtin_amdgcn_perm((u32)hi[t][i], (u32)(lo[t][i] + hi[t][i]), 0x05040100u);
        }
      }
    };
    auto run = [&](auto nts) {
      constexpr int NTS = decltype(nts)::value;
      constexpr int SPS = (RI_MAXPI + NTS - 1) / NTS;      // pieces per sub-step of the diagonal block
      auto diag = [&](int u) {
        if (!have_prev) return;
        if (u == 0) STAMP(2);
        if (u * SPS < RI_MAXPI) drain(u * SPS, (u + 1) * SPS < RI_MAXPI ? (u + 1) * SPS : RI_MAXPI, lane);
        if (u == NTS - 1) STAMP(3);
      };
      auto epi = [&](auto &lo, auto &hi) {
        finish(lane);
        image(lo, hi);
        STAMP(7);
      };
      toeplitz_strip<M_ENC, NTS>(st0, st0, tb0, tb1, g, kb0, mlow, epi, 0, 0, 0x7fffffff, NoPause(), diag);
    };
    switch (nt) {
      case 0:
        if (have_prev) drain(0, RI_MAXPI, lane);
        finish(lane);
        break;
      case 1: run(std::integral_constant<int, 1>{}); break;
      case 2: run(std::integral_constant<int, 2>{}); break;
      case 3: run(std::integral_constant<int, 3>{}); break;
      default: run(std::integral_constant<int, 4>{}); break;
    }
    if (rb_next < nrb) stage_r(rb_next, lane);
    STAMP(8);
    prev = cur;
    have_prev = true;
  }
  ri_barrier();                                          // the last row block's image
  if (have_prev) drain(0, RI_MAXPI, lane0);
}

NTRU_STAMPS_READER(ntru_debug_read_stamps_rowimage)

#endif   // NTRU_EXPERIMENTS

// ---- encryptBits + packOutput(q - 1, N, e) in one kernel (index.js:87-110, :572-596) ---------------------------
// The same row-image kernel with a different drain: what leaves the CU is the BN254 field-element form of e alone -- PER = 252 / BITS
// fields of BITS bits per 32-byte element, os elements per row, the row block's 32 os elements one contiguous, 32-byte aligned run.
// One element per lane and drain instruction: its PER raw dwords are PER consecutive dwords of the image (lane stride PER dwords: odd
// or 2 mod 4, no two lanes of a quarter wave in one bank for PER = 21, 19; two-way for 22), its plaintext bytes 24 consecutive bytes
// of m (seven aligned dwords) requested a whole loop ahead; e = (low + high + m) mod q is formed per field and shifted into eight dwords at compile-time
// positions.  No quotientE, no partial pieces, no 2-byte stores: 32 os bytes per row instead of 4 N.
constexpr int RP_MAXPI = 3;                      // drain instructions per wave: 8 * 3 * 64 = 1536 elements >= 32 os (os <= 48: every N whose image fits the LDS)

template <int BITS>
__global__ __launch_bounds__(RI_THREADS, 2) void k_encrypt_wp(MGeom g, u32 q, const u16 *__restrict__ h, const uint8_t *__restrict__ r,
                                                              const uint8_t *__restrict__ m, long B, int os,
                                                              unsigned long long *__restrict__ packed) {
  constexpr int PER = 252 / BITS;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  u32 *T0 = (u32 *)lds, *T1 = T0 + 4 * g.tpitch;
  unsigned char *stA = (unsigned char *)(T1 + 4 * g.tpitch);
  unsigned char *img = stA + 32 * g.pitchA;
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int hthr = (int)(q >> 1) - 65;
  auto hs_of = [&](int i) { int hv = (int)(h[i] & (q - 1)); return hv > hthr ? hv - (int)q : hv; };
  build_toeplitz_array(T0, g, [&](int i) { const int hs = hs_of(i); return ((hs + 64) & 127) - 64; }, tid0, RI_THREADS);
  build_toeplitz_array(T1, g, [&](int i) { const int hs = hs_of(i); const int d0 = ((hs + 64) & 127) - 64; return ((hs - d0) >> 7) * 4; }, tid0, RI_THREADS);
  const long nrb = (B + 31) >> 5;
  const int sbase = g.NT / RI_WAVES, srem = g.NT % RI_WAVES;
  const int nt = sbase + (wave < srem ? 1 : 0), kb0 = wave * sbase + (wave < srem ? wave : srem);
  const int c0 = __builtin_amdgcn_readfirstlane((int)((unsigned long long)m & 3));
  const u32 os_inv = 0xFFFFFFFFu / (u32)os + 1u;             // t / os = umulhi(t, os_inv) for t < 2^16
  constexpr int RPW = 32 / RI_WAVES;
  RawChunks<1> in_r[RPW];
  auto request_r = [&](long rb, int lane) {
    const long b0 = rb << 5 < B ? rb << 5 : B;
    const AlignedSrc src_r = aligned_src(r + b0 * g.N, (B - b0) * g.N);
#pragma unroll
    for (int j = 0; j < RPW; j++) in_r[j] = load_raw<1>(src_r, src_r.a0 + (wave + RI_WAVES * j) * g.N + 16 * lane, 0);
  };
  auto stage_r = [&](long rb, int lane) {
    const long b0 = rb << 5 < B ? rb << 5 : B;
    const int ar = (int)((unsigned long long)(r + b0 * g.N) & 15);
    const v4i mk = col_mask16(16 * lane, g.N);
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int row = wave + RI_WAVES * j;
      v4i v[1];
      shift_raw<1>(in_r[j], ar + row * g.N, v);
      if (lane < 2 * g.NT) *(v4i *)(stA + row * g.pitchA + 16 * lane) = v[0] & mk;
    }
  };
  struct Prev { long b0; int rows; };
  // element t = 64 (8 j + wave) + lane of a row block: (row, k) = (t / os, t % os); its fields are columns PER k .. PER k + PER - 1
  auto task = [&](int j, int lane, int rows, int *row, int *col0) {
    const u32 t = 64u * (u32)(RI_WAVES * j + wave) + (u32)lane;
    const u32 rw = __umulhi(t, os_inv);
    *row = (int)rw; *col0 = PER * (int)(t - rw * (u32)os);
    return (int)rw < rows;
  };
  // Plaintext bytes of an element: PER <= 22 consecutive bytes of m at any alignment = 7 ALIGNED dwords, each range-checked on its own
  // against the row block's bytes rounded up to a dword (a wider load that straddles the end of m comes back as zero altogether, and
  // nothing may be read past the last dword of the array); the byte phase is taken out at drain time.
  u32 mreg[RP_MAXPI][7];
  auto request_m = [&](const Prev &pv, int lane) {
    const __amdgpu_buffer_rsrc_t rsm = rows_rsrc((const void *)((unsigned long long)(m + pv.b0 * g.N) & ~3ULL), (c0 + pv.rows * g.N + 3) & ~3);
#pragma unroll
    for (int j = 0; j < RP_MAXPI; j++) {
      int row, col0;
      const bool ok = task(j, lane, pv.rows, &row, &col0);
      const int off = (c0 + row * g.N + col0) & ~3;
#pragma unroll
      for (int c = 0; c < 7; c++) {
        mreg[j][c] = 0;
        if (64 * (RI_WAVES * j + wave) < 32 * os) mreg[j][c] = (u32)__builtin_amdgcn_raw_buffer_load_b32(rsm, ok ? off + 4 * c : (int)0x80000000, 0, 0);
      }
    }
  };
  Prev prev = {0, 0};
  bool have_prev = false;
  auto drain = [&](int j0, int j1, int lane) {
    Prev pv = prev;
    asm volatile("" : "+s"(pv.b0), "+s"(pv.rows));
    const __amdgpu_buffer_rsrc_t rsP = rows_rsrc((const unsigned char *)(packed + pv.b0 * os * 4), pv.rows * os * 32);
#pragma unroll
    for (int j = 0; j < RP_MAXPI; j++)
      if (j >= j0 && j < j1 && 64 * (RI_WAVES * j + wave) < 32 * os) {
        int row, col0;
        const bool ok = task(j, lane, pv.rows, &row, &col0);
        const u32 *px = (const u32 *)img + (ok ? row * g.N + col0 : 0);       // (a row's last element reads up to PER - 1 dwords past it: the pad)
        u32 x[PER];
#pragma unroll
        for (int f = 0; f < PER; f++) x[f] = px[f];
        const u32 ph = 8u * (u32)((c0 + row * g.N + col0) & 3);             // bit phase of the element's first plaintext byte in its dword
        u32 mw[6];
#pragma unroll
        for (int c = 0; c < 6; c++) mw[c] = __builtin_amdgcn_alignbit(mreg[j][c + 1], mreg[j][c], ph);
        u32 o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int f = 0; f < PER; f++) {
          const u32 mb = (mw[f >> 2] >> (8 * (f & 3))) & 0xFFu;
          u32 ev = (x[f] + mb) & (q - 1);
          ev = col0 + f < g.N ? ev : 0u;
          const int pos = BITS * f, d = pos >> 5, sh = pos & 31;
          o[d] |= ev << sh;
          if (sh + BITS > 32) o[d + 1] |= ev >> (32 - sh);
        }
        const int voff = ok ? 32 * (int)(64 * (RI_WAVES * j + wave) + lane) : (int)0x80000000;
        __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)o[0], (int)o[1], (int)o[2], (int)o[3]}, rsP, voff, 0, ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128((v4i){(int)o[4], (int)o[5], (int)o[6], (int)o[7]}, rsP, ok ? voff + 16 : (int)0x80000000, 0, ST_AUX);
      }
  };

  long rb = blockIdx.x;
  if (rb < nrb) { request_r(rb, lane0); stage_r(rb, lane0); }
  for (; rb < nrb; rb += gridDim.x) {
    int lane = lane0, N = g.N;
    asm volatile("" : "+v"(lane), "+s"(N));
    ri_barrier();                                        // B: stage of rb and image of `prev` complete (first pass: key arrays)
    const long rb_next = rb + gridDim.x;
    if (rb_next < nrb) request_r(rb_next, lane);
    const u32 *tb0 = frag_lane_base(T0, g, lane), *tb1 = frag_lane_base(T1, g, lane);
    const unsigned char *st0 = stA + (lane & 31) * g.pitchA + 16 * (lane >> 5);
    u32 mlow[4];
    diag_low_mask(lane, mlow);
    const long b0 = rb << 5;
    const Prev cur = {b0, (int)(B - b0 < 32 ? B - b0 : 32)};
    auto finish = [&](int lane_) {                           // after the loops
      request_m(cur, lane_);                               // the plaintext bytes the NEXT loops will add while draining this row block
      ri_barrier();                                      // A: every wave has left its loops -- stage free, image drained
    };
    auto image = [&](auto &lo, auto &hi) {
      constexpr int NTS = sizeof(lo) / sizeof(lo[0]);
      const int colb = 32 * kb0 + (lane & 31), rowb = 4 * (lane >> 5);
#pragma unroll
      for (int t = 0; t < NTS; t++) {
        if (colb + 32 * t < N) {
          u32 *pe = (u32 *)img + rowb * N + colb + 32 * t;
#pragma unroll
          for (int i = 0; i < 16; i++) pe[((i & 3) + 8 * (i >> 2)) * N] = (u32)(lo[t][i] + hi[t][i]);     // (only the low 16 bits are used)
        }
      }
    };
    auto run = [&](auto nts) {
      constexpr int NTS = decltype(nts)::value;
      constexpr int SPS = (RP_MAXPI + NTS - 1) / NTS;      // drain instructions per sub-step of the diagonal block
      auto diag = [&](int u) {
        if (!have_prev) return;
        if (u * SPS < RP_MAXPI) drain(u * SPS, (u + 1) * SPS < RP_MAXPI ? (u + 1) * SPS : RP_MAXPI, lane);
      };
      auto epi = [&](auto &lo, auto &hi) {
        finish(lane);
        image(lo, hi);
      };
      toeplitz_strip<M_ENC, NTS>(st0, st0, tb0, tb1, g, kb0, mlow, epi, 0, 0, 0x7fffffff, NoPause(), diag);
    };
    switch (nt) {
      case 0:
        if (have_prev) drain(0, RP_MAXPI, lane);
        finish(lane);
        break;
      case 1: run(std::integral_constant<int, 1>{}); break;
      case 2: run(std::integral_constant<int, 2>{}); break;
      case 3: run(std::integral_constant<int, 3>{}); break;
      default: run(std::integral_constant<int, 4>{}); break;
    }
    if (rb_next < nrb) stage_r(rb_next, lane);
    prev = cur;
    have_prev = true;
  }
  ri_barrier();                                          // the last row block's image
  if (have_prev) drain(0, RP_MAXPI, lane0);
}

// ---- host side ----------------------------------------------------------------------------------------------------------
// Kernel path 10 of a -DNTRU_EXPERIMENTS library: dense rows, 2-byte aligned result arrays, N <= 1024 columns in 8 strips
// of <= 4 tiles, images + stage + key arrays within 160 KB of LDS.
int ntru_launch_encrypt_rowimage(ntru_engine *eng, int N, int q, int ld, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m,
                                 int64_t B, uint16_t *d_e, uint16_t *d_quotE) {
#ifndef NTRU_EXPERIMENTS
  return NTRU_NOT_TAKEN;
#else
  MGeom mg;
  if (ld != N || !make_mgeom(eng, N, q, ld, &mg)) return NTRU_NOT_TAKEN;
  if (((uintptr_t)d_e & 1) != 0 || (d_quotE && (((uintptr_t)d_e ^ (uintptr_t)d_quotE) & 15) != 0) || mg.NT > 4 * RI_WAVES) return NTRU_NOT_TAKEN;
  const size_t lds = (size_t)32 * mg.tpitch + (size_t)32 * mg.pitchA + (size_t)ri_img_bytes(N);
  if (lds > 160 * 1024 || 64 * N + 16 > 1024 * RI_WAVES * RI_MAXPI) return NTRU_NOT_TAKEN;
  const long nrb = (long)((B + 31) / 32);
  dim3 grid;
  if (int rc = resident_grid(eng, k_encrypt_w, lds, nrb, &grid, RI_THREADS)) return rc;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_w");
  hipLaunchKernelGGL(k_encrypt_w, grid, dim3(RI_THREADS), lds, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, d_e, d_quotE);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
#endif
}

// encryptBits + packOutput(q - 1, N, e) fused (default library): dense rows, q in {2048, 4096, 8192} (11 / 12 / 13-bit fields), N <= 1024
// columns in 8 strips of <= 4 tiles, 32-byte aligned packed rows (16-byte aligned d_packed), image + stage + key arrays within 160 KB.
int ntru_launch_encrypt_pack_rowimage(ntru_engine *eng, int N, int q, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m,
                                      int64_t B, uint64_t *d_packed, int os) {
  MGeom mg;
  if (!make_mgeom(eng, N, q, N, &mg)) return NTRU_NOT_TAKEN;
  if (q != 2048 && q != 4096 && q != 8192) return NTRU_NOT_TAKEN;
  if (((uintptr_t)d_packed & 15) != 0 || mg.NT > 4 * RI_WAVES || 32 * os > 64 * RI_WAVES * RP_MAXPI) return NTRU_NOT_TAKEN;
  const size_t lds = (size_t)32 * mg.tpitch + (size_t)32 * mg.pitchA + (size_t)ri_img_bytes(N) + 128;     // + the pad a row's last element reads
  if (lds > 160 * 1024) return NTRU_NOT_TAKEN;
  const long nrb = (long)((B + 31) / 32);
  dim3 grid;
  auto go = [&](auto kern) -> int {
    if (int rc = resident_grid(eng, kern, lds, nrb, &grid, RI_THREADS)) return rc;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_wp");
    hipLaunchKernelGGL(kern, grid, dim3(RI_THREADS), lds, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, os, (unsigned long long *)d_packed);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  };
  return q == 2048 ? go(k_encrypt_wp<11>) : q == 4096 ? go(k_encrypt_wp<12>) : go(k_encrypt_wp<13>);
}
