// matrix_decrypt.hip -- MI355X (gfx950), family 4: decryptBits (index.js:111-140) for a batch under ONE shared key: a = f * e mod q,
// split, centred lift (index.js:117 verbatim), c = fp * b mod p, split -- both products as batch x Toeplitz matrix products on the int8
// matrix cores, chained in one kernel (the lifted message never leaves the CU).  matrix_common.h holds the tile geometry and the
// strip loop; tools/mfma_model.py is the executable specification.
#include "matrix_common.h"

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

// FP4: product 2 (ternary x ternary) on the block-scaled fp4 matrix instruction, K = 64 per instruction (matrix_common.h,
// toeplitz_strip_fp4): the lifted message is expanded into a NIBBLE stage, fp's key array is a nibble stream in 8 shifted copies.
#ifndef NTRU_EXPERIMENTS       // the fp4 machinery lives in the experiments build only: inert stand-ins so that decrypt_m_body<.., false> compiles
struct FGeom { int NT2, pitch4, tp4; };
static __host__ __device__ inline FGeom make_fgeom(int) { return FGeom{0, 0, 0}; }
template <class D> static __device__ __forceinline__ void build_toeplitz_array4(u32 *, const MGeom &, const FGeom &, D, int, int) {}
static __device__ __forceinline__ const u32 *frag4_lane_base(const u32 *T4, const MGeom &, const FGeom &, int) { return T4; }
template <int NT_S, class Epi>
static __device__ __forceinline__ void toeplitz_strip_fp4(const unsigned char *, const u32 *, const FGeom &, int, int, Epi) {}
#endif
// PACK (k_decrypt_mp): packOutput(p - 1, N, value) (index.js:572-596: 2 bits per value, 126 values per 252-bit field element, four
// little-endian 64-bit limbs per element) comes out of the same kernel: product 2's epilogue drops its values into a 2-bit packed
// LDS image [8 row groups][columns] (4 rows per byte -- the layout product 1 uses for the lifted message: one ds_write_b8 per four
// accumulator registers), and at the top of the next trip every thread turns 16 columns of one row into one dword of the packed
// row (the row block's packed rows are contiguous: consecutive threads store consecutive dwords).  `value` itself is optional
// then: the pipeline's pack mode writes 32 ceil(N / 126) bytes per item instead of N + 32 ceil(N / 126).
// (First version: DPP gather + ds_or_b32 into an image of the packed rows: bit-exact, 3.65 ms per 2^20 against 1.83 ms for the
// plain value-only kernel -- 1248 LDS atomics per row block with 4 active lanes each.)
template <int GROUPS, bool DMA = false, bool FP4 = false, bool PACK = false>
static __device__ __forceinline__ void decrypt_m_body(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                      const uint8_t *__restrict__ fp,
                                                      const u16 *__restrict__ e, long B,
                                                      uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                      u16 *__restrict__ rem1, uint8_t *__restrict__ quot2,
                                                      unsigned long long *__restrict__ packed = nullptr, int pack_os = 0) {
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
  static_assert(!(DMA && FP4), "the fp4 second product is built on the register-staged kernels");
  static_assert(!PACK || (GROUPS == 1 && !DMA && !FP4), "fused packOutput is built on the plain four-wave kernel");
  const FGeom f4 = make_fgeom(g.NT);
  const int RP = DMA ? dec_dma_row_pitch(g.NT) : g.pitchA;                       // row pitch of the operand stage(s)
  const int m3b = DMA ? dec_dma_m3_bytes(g.N, (int)p) : 0;
  const int gbytes = (DMA ? 32 * RP : 64 * g.pitchA) + 256 * g.NT;
  unsigned char *stHi = DMA ? lds + m3b + group * gbytes + g.pitchA : lds + group * gbytes;   // DMA: the high plane of row R at slot R + pitchA
  unsigned char *stLo = DMA ? lds + m3b + group * gbytes : stHi + 32 * g.pitchA;
  unsigned char *blp = DMA ? stLo + 32 * RP : stLo + 32 * g.pitchA;              // [8 row groups][32 NT columns]: 4 rows x 2 bits per byte
  u32 *TF = (u32 *)(lds + m3b + GROUPS * gbytes), *TP = TF + 4 * g.tpitch;
  unsigned char *lift_lut = (unsigned char *)(TP + (FP4 ? 8 * f4.tp4 : 4 * g.tpitch));   // [q]: centred lift followed by mod p, index.js:117 verbatim
  // PACK: 2-bit image of product 2's values, [8 row groups][pcols] bytes, pcols = 126 pack_os + 16 >= 32 NT (zero beyond N).  It lives
  // in the e_hi stage behind the mod-p tables (dead between product 1's last loop and the next trip's staging: no LDS of its own --
  // 7 KB more would cost the second workgroup of the CU); it is turned into packed dwords at the top of the next trip, and a barrier
  // separates that from the staging writes.
  const int pcols = 126 * pack_os + 16, pimg_bytes = (8 * pcols + 15) & ~15;
  unsigned char *pimg = stHi + ((((int)((p - 1) * (p - 1)) * g.N + 4) & ~3) + (int)((p - 1) * (p - 1)) * g.N + 1 + 15 & ~15);
  auto pack_flush = [&](long rbx) {                      // the image of row block rbx -> packed[32 rbx ..]; thread = one dword of one row
    const long b0x = rbx << 5;
    const int rows_left = (int)(B - b0x < 32 ? B - b0x : 32);
    u32 *dst = (u32 *)packed + b0x * 8 * pack_os;
    const int per_row = 8 * pack_os;
    for (int x = (int)threadIdx.x; x < rows_left * per_row; x += BLOCK_THREADS) {
      const int R = x / per_row, dw = x - R * per_row, o = dw >> 3, k = dw & 7;
      const unsigned char *src = pimg + (2 * (R >> 3) + ((R >> 2) & 1)) * pcols + 126 * o + 16 * k;
      const int sh = 2 * (R & 3);
      u32 out = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        u32 d;                                               // four columns; the address is only 2-byte aligned
        __builtin_memcpy(&d, src + 4 * c, 4);
        u32 t = (d >> sh) & 0x03030303u;
        t |= t >> 6;
        t = (t | (t >> 12)) & 0xFFu;
        out |= t << (8 * c);
      }
      dst[x] = k == 7 ? out & 0x0FFFFFFFu : out;             // 126 = 7 x 16 + 14: the eighth dword of an element holds 14 values
    }
  };
  auto pack_wipe = [&]() {                               // the staging of product 1 has been here since: columns >= N must read zero again
    for (int x = (int)threadIdx.x; x < pimg_bytes / 16; x += BLOCK_THREADS) *(uint4 *)(pimg + 16 * x) = make_uint4(0u, 0u, 0u, 0u);
  };
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
  if (FP4) build_toeplitz_array4(TP, g, f4, [&](int i) { return (int)fp[i]; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
  else build_toeplitz_array(TP, g, [&](int i) { return (int)fp[i]; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
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
    const u32 *tbf = frag_lane_base(TF, g, lane), *tbp = FP4 ? nullptr : frag_lane_base(TP, g, lane);

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
        raw[j] = load_raw<2>(src_e, pos0 + 32 * lane, sh[j]);
      }
    };
    wg_barrier();
    STAMP(1);
    if (!DMA) request_rows();
    if (PACK && it > 0) {                                // every wave has left the previous trip's epilogues; this trip's rows are in flight
      pack_flush((long)blockIdx.x * GROUPS + group + (it - 1) * stride);
      wg_barrier();                                      // ... before the staging below overwrites the image
    }
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
                {
                  if (decltype(wr)::value) __builtin_amdgcn_raw_buffer_store_b16((u16)x, rs_r1, voff[t], so, ST_AUX);
                  if (decltype(wq)::value) __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi[t][i]) & (q - 1)), rs_q1, voff[t], so, ST_AUX);
                }
              }
            }
#pragma unroll
            for (int t = 0; t < NTS; t++)
#pragma unroll
              for (int ii = 0; ii < 4; ii++) lv[t][ii] = lift_lut[xs[t][ii]];
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
    if (PACK) pack_wipe();
    if (!DMA) for (int x = tid0; x <= (int)((p - 1) * (p - 1)) * N; x += BLOCK_THREADS) {
      const u32 rm = mod_small((u32)x, p);
      m3_lut[x] = (unsigned char)(rm ? p - rm : 0u);
      m3_lut[M3V + x] = (unsigned char)rm;
    }
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
      if (!FP4) {
#pragma unroll
        for (int j = 0; j < RPW; j++) {
          const int row = wave + WAVES_PER_BLOCK * j, sh = 2 * (row & 3);
#pragma unroll
          for (int it = 0; it < 4; it++)
            if (lane + 64 * it < 8 * g.NT) *(u32 *)(stLo + row * RP + 4 * (lane + 64 * it)) = (pv[j][it] >> sh) & 0x03030303u;
        }
      } else {
        // nibble stage: 8 columns per dword, fp4 codes (value << 1).  pv[.][it] of lane l is the packed dword u = 64 it + l = columns
        // 4 u .. 4 u + 3 of four rows; the dword of the four columns after them sits in lane l + 1 (in pv[.][it + 1] of lane 0 for
        // lane 63).  Even lanes write dword u / 2 of the row; what lies beyond the last tile (half a block when NT is odd) is zero.
#pragma unroll
        for (int j = 0; j < RPW; j++) {
          const int row = wave + WAVES_PER_BLOCK * j, sh = 2 * (row & 3);
#pragma unroll
          for (int it = 0; it < 4; it++) {
            const int u = lane + 64 * it;
            const u32 own = u < 8 * g.NT ? pv[j][it] : 0u;
            const u32 nx = it < 3 ? pv[j][it + 1] : 0u;
            const u32 up = (u32)__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, (int)pv[j][it]);      // lane l + 1's dword
            const u32 first = (u32)__builtin_amdgcn_readfirstlane((int)nx);                                // lane 0's next dword
            const u32 d1 = u + 1 < 8 * g.NT ? (lane == 63 ? first : up) : 0u;
            const u32 v0 = (own >> sh) & 0x03030303u, v1 = (d1 >> sh) & 0x03030303u;
            const u32 n0 = v0 | (v0 >> 4), n1 = v1 | (v1 >> 4);       // bytes 0 and 2: (c0 | c1 << 4), (c2 | c3 << 4)
            const u32 nib = __builtin_amdgcn_perm(n1, n0, 0x06040200u) << 1;
            if ((lane & 1) == 0 && u < 16 * f4.NT2) *(u32 *)(stLo + row * f4.pitch4 + 2 * u) = nib;
          }
        }
      }
    }
    STAMP(9);
    wg_barrier();
    STAMP(10);
    // ---- product 2: c = fp * lifted mod p
    sidx = 0;
    int lane2 = lane0;                                   // FP4: this lane's nibble-stage row and fragment base, made here (held from the
    asm volatile("" : "+v"(lane2));                      // top of the trip they cost four registers the first product has no room for)
    const unsigned char *st4 = stLo + (lane2 & 31) * f4.pitch4 + 16 * (lane2 >> 5);
    const u32 *tbp4 = frag4_lane_base(TP, g, f4, lane2);
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
        const __amdgpu_buffer_rsrc_t rs_v = rows_rsrc(value ? value + bb * LD : nullptr, value ? lf : 0);
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
                va[t][ii] = m3_lut[(u32)(lo[t][4 * j + ii] + hi[t][4 * j + ii] + M3V)];
                vb[t][ii] = decltype(wq)::value ? (u32)m3_lut[(u32)hi[t][4 * j + ii]] : 0u;
              }
            if (PACK) {
#pragma unroll
              for (int t = 0; t < NTS; t++) {
                const int col = 32 * (kb0 + t) + (lane & 31);
                const u32 pk = va[t][0] | (va[t][1] << 2) | (va[t][2] << 4) | (va[t][3] << 6);
                if (col < N) pimg[((lane >> 5) + 2 * j) * pcols + col] = (unsigned char)pk;   // [row group 2j+hh][column]; columns >= N stay zero
              }
            }
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
              if (PACK && value == nullptr) break;         // pack mode without the plain values
#pragma unroll
              for (int t = 0; t < NTS; t++) {
                const int so = (ii + 8 * j) * LD + 32 * (kb0 + t);
                {
                  __builtin_amdgcn_raw_buffer_store_b8((uint8_t)va[t][ii], rs_v, voff[t], so, ST_AUX);
                  if (decltype(wq)::value) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)vb[t][ii], rs_q2, voff[t], so, ST_AUX);
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
        case 1: if (FP4) toeplitz_strip_fp4<1>(st4, tbp4, f4, kb0, lane, epi); else toeplitz_strip<M_DEC2, 1>(st0, st0, tbp, tbp, g, kb0, mlow, epi, stamp_iter, 11 + 2 * sidx); break;
        case 2: if (FP4) toeplitz_strip_fp4<2>(st4, tbp4, f4, kb0, lane, epi); else toeplitz_strip<M_DEC2, 2>(st0, st0, tbp, tbp, g, kb0, mlow, epi, stamp_iter, 11 + 2 * sidx); break;
        case 3: if (FP4) toeplitz_strip_fp4<3>(st4, tbp4, f4, kb0, lane, epi); else toeplitz_strip<M_DEC2, 3>(st0, st0, tbp, tbp, g, kb0, mlow, epi, stamp_iter, 11 + 2 * sidx); break;
        default: if (FP4) toeplitz_strip_fp4<4>(st4, tbp4, f4, kb0, lane, epi); else toeplitz_strip<M_DEC2, 4>(st0, st0, tbp, tbp, g, kb0, mlow, epi, stamp_iter, 11 + 2 * sidx); break;
      }
      sidx++;
      if (sidx < rounds) phase(8);
    }, GROUPS == 2);
  }
  if (GROUPS == 2 && group == 0) wg_barrier();        // group 1's last phase
  if (PACK && iters > 0) {
    wg_barrier();                                       // the last trip's epilogues
    const long last = (long)blockIdx.x + (iters - 1) * stride;
    if (last < nrb) pack_flush(last);
  }
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

// decryptBits + packOutput(p - 1, N, value) in one kernel (decrypt_m_body<.., PACK>); `value` may be NULL.
__global__ __launch_bounds__(BLOCK_THREADS, 2) void k_decrypt_mp(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                              const uint8_t *__restrict__ fp,
                                                              const u16 *__restrict__ e, long B,
                                                              uint8_t *__restrict__ value, unsigned long long *__restrict__ packed, int pack_os) {
  decrypt_m_body<1, false, false, true>(g, q, p, f, fp, e, B, value, nullptr, nullptr, nullptr, packed, pack_os);
}

#ifdef NTRU_EXPERIMENTS
// The same two kernels with product 2 on the fp4 matrix instruction (kernel path 11): bit-exact, 5 % slower (profiles/r04_fp4_product2.txt).
__global__ __launch_bounds__(BLOCK_THREADS, 2) void k_decrypt_mq(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                              const uint8_t *__restrict__ fp,
                                                              const u16 *__restrict__ e, long B,
                                                              uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                              u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  decrypt_m_body<1, false, true>(g, q, p, f, fp, e, B, value, quot1, rem1, quot2);
}

__global__ __launch_bounds__(2 * BLOCK_THREADS, 1) void k_decrypt_m8q(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                                   const uint8_t *__restrict__ fp,
                                                                   const u16 *__restrict__ e, long B,
                                                                   uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                                   u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  decrypt_m_body<2, false, true>(g, q, p, f, fp, e, B, value, quot1, rem1, quot2);
}

__global__ __launch_bounds__(2 * BLOCK_THREADS, 1) void k_decrypt_m8d(MGeom g, u32 q, u32 p, const int8_t *__restrict__ f,
                                                                   const uint8_t *__restrict__ fp,
                                                                   const u16 *__restrict__ e, long B,
                                                                   uint8_t *__restrict__ value, u16 *__restrict__ quot1,
                                                                   u16 *__restrict__ rem1, uint8_t *__restrict__ quot2) {
  decrypt_m_body<2, true>(g, q, p, f, fp, e, B, value, quot1, rem1, quot2);
}
#endif

NTRU_STAMPS_READER(ntru_debug_read_stamps_dec)

// ---- host side ----------------------------------------------------------------------------------------------------------
// decryptBits + packOutput fused (k_decrypt_mp): p == 3, the matrix path's range, a 16-byte aligned packed array.
int ntru_launch_decrypt_pack_matrix(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const uint8_t *d_fp, const uint16_t *d_e,
                                    int64_t B, uint8_t *d_value, uint64_t *d_packed, int out_size) {
  MGeom mg;
  if (p != 3 || ((uintptr_t)d_packed & 15) != 0 || !make_mgeom(eng, N, q, N, &mg)) return NTRU_NOT_TAKEN;
  const size_t lds = (size_t)32 * mg.tpitch + (size_t)64 * mg.pitchA + (size_t)256 * mg.NT + (((size_t)q + 15) & ~(size_t)15);
  // the 2-bit image of the values lives in the e_hi stage behind the mod-p tables
  const size_t m3_end = ((((size_t)4 * N + 4) & ~(size_t)3) + (size_t)4 * N + 1 + 15) & ~(size_t)15, img = ((size_t)8 * (126 * out_size + 16) + 15) & ~(size_t)15;
  if (lds > 160 * 1024 || m3_end + img > (size_t)32 * mg.pitchA || 126 * out_size + 16 < 32 * mg.NT) return NTRU_NOT_TAKEN;
  const long nrb = (long)((B + 31) / 32);
  dim3 grid;
  if (int rc = resident_grid(eng, k_decrypt_mp, lds, nrb, &grid)) return rc;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_mp");
  hipLaunchKernelGGL(k_decrypt_mp, grid, dim3(BLOCK_THREADS), lds, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e, (long)B, d_value,
                     (unsigned long long *)d_packed, out_size);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

// Kernel paths: 4 -> k_decrypt_m (two free-running workgroups per CU); 5 -> k_decrypt_m8 (one workgroup of two lock-step groups)
// wherever its LDS fits; 0 = auto -> k_decrypt_m8 where a product takes two rounds of strips (N > 512) and every witness array is
// asked for: 2.52 against 2.63 ms per 2^20 at N = 821 (profiles/archive/r02_ab_lockstep_phase_masks.txt), 2.16 against 2.24 ms at N = 701;
// at N = 509 (one round) it is 6 % slower, and so it is without the witness arrays (shorter epilogues: 2.23 against 2.02 ms).
// -DNTRU_EXPERIMENTS builds add 8 (k_decrypt_m8d: 5 with direct-to-LDS loads of the next row block; 1-3 % slower).
int ntru_launch_decrypt_matrix(ntru_engine *eng, int N, int q, int p, int ld, const int8_t *d_f, const uint8_t *d_fp, const uint16_t *d_e,
                               int64_t B, uint8_t *d_value, uint16_t *d_quot1, uint16_t *d_rem1, uint8_t *d_quot2) {
  MGeom mg;
  if (p != 3 || !make_mgeom(eng, N, q, ld, &mg)) return NTRU_NOT_TAKEN;
  const size_t lds = (size_t)32 * mg.tpitch + (size_t)64 * mg.pitchA + (size_t)256 * mg.NT + (((size_t)q + 15) & ~(size_t)15);
  const long nrb = (long)((B + 31) / 32);
  dim3 grid;
#ifdef NTRU_EXPERIMENTS
  if (eng->path == 11) {                                 // product 2 on the fp4 matrix instruction (K = 64): fp's array is 8 nibble-shifted copies
    const size_t keys = (size_t)16 * mg.tpitch + fp4_array_bytes(mg.NT);
    const size_t lut = ((size_t)q + 15) & ~(size_t)15, per_group = (size_t)64 * mg.pitchA + (size_t)256 * mg.NT;
    const bool lockstep = mg.NT > 16 && d_quot1 && d_rem1 && d_quot2 && keys + lut + 2 * per_group <= 160 * 1024;
    if (lockstep) {
      const size_t l8 = keys + lut + 2 * per_group;
      if (int rc = resident_grid(eng, k_decrypt_m8q, l8, (nrb + 1) / 2, &grid, 2 * BLOCK_THREADS)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_m8q");
      hipLaunchKernelGGL(k_decrypt_m8q, grid, dim3(2 * BLOCK_THREADS), l8, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e,
                         (long)B, d_value, d_quot1, d_rem1, d_quot2);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
    const size_t l4 = keys + lut + per_group;
    if (l4 <= 160 * 1024) {
      if (int rc = resident_grid(eng, k_decrypt_mq, l4, nrb, &grid)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_mq");
      hipLaunchKernelGGL(k_decrypt_mq, grid, dim3(BLOCK_THREADS), l4, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e,
                         (long)B, d_value, d_quot1, d_rem1, d_quot2);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
  }
  if (eng->path == 8) {                                  // lock-step + direct-to-LDS loads of the next row block
    const size_t ldsd = (size_t)dec_dma_m3_bytes(N, p) + 2 * ((size_t)32 * dec_dma_row_pitch(mg.NT) + (size_t)256 * mg.NT) +
                        (size_t)32 * mg.tpitch + (((size_t)q + 15) & ~(size_t)15);
    if (ldsd <= 160 * 1024) {
      if (int rc = resident_grid(eng, k_decrypt_m8d, ldsd, (nrb + 1) / 2, &grid, 2 * BLOCK_THREADS)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_m8d");
      hipLaunchKernelGGL(k_decrypt_m8d, grid, dim3(2 * BLOCK_THREADS), ldsd, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e,
                         (long)B, d_value, d_quot1, d_rem1, d_quot2);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
  }
#endif
  if (eng->path == 5 || eng->path == 9 || (eng->path == 0 && mg.NT > 16 && d_quot1 && d_rem1 && d_quot2)) {
    const size_t lds8 = 2 * ((size_t)64 * mg.pitchA + (size_t)256 * mg.NT) + (size_t)32 * mg.tpitch + (((size_t)q + 15) & ~(size_t)15);
    if (lds8 <= 160 * 1024) {
      if (int rc = resident_grid(eng, k_decrypt_m8, lds8, (nrb + 1) / 2, &grid, 2 * BLOCK_THREADS)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_m8");
      hipLaunchKernelGGL(k_decrypt_m8, grid, dim3(2 * BLOCK_THREADS), lds8, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e,
                         (long)B, d_value, d_quot1, d_rem1, d_quot2);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
  }
  if (lds > 160 * 1024) return NTRU_NOT_TAKEN;
  if (int rc = resident_grid(eng, k_decrypt_m, lds, nrb, &grid)) return rc;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_decrypt_m");
  hipLaunchKernelGGL(k_decrypt_m, grid, dim3(BLOCK_THREADS), lds, eng->stream, mg, (u32)q, (u32)p, d_f, d_fp, d_e,
                     (long)B, d_value, d_quot1, d_rem1, d_quot2);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}
