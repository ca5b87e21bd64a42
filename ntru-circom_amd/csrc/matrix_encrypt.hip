// matrix_encrypt.hip -- MI355X (gfx950), family 4: encryptBits (index.js:87-110) for a batch under ONE shared key as batch x Toeplitz
// matrix products on the int8 matrix cores (v_mfma_i32_32x32x32_i8, int32 accumulation: exact), the closed-form split by 1 - x^N
// (index.js:358-401) and the plaintext add (index.js:235-244) fused into the epilogue.  matrix_common.h holds the
// tile geometry and the strip loop; tools/mfma_model.py is the executable specification.
#include "matrix_common.h"

// encryptBits on the matrix cores: e = (r * h + m) split by 1 - x^N; r in {0..3} bytes, h < q <= 8192.
// h is taken in the representative hs = d0 + 128 d1, d0 in [-64,63], 4 d1 in [-128,124]; planes [r | 32 r] x [d0 ; 4 d1].
// MAXT: widest strip.  8 (one workgroup per CU, 512 registers per wave, one strip per wave and row block) was measured at
// 2.28 ms per 2^20 against 1.51-1.58 ms for 4: with one wave per SIMD nothing overlaps the matrix loops
// (EXPERIMENTS.md, round 2); only 4 is instantiated.
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
  // Digits of h: value = d0 + 128 d1 against the planes [r | 32 r] x [d0 ; 4 d1].  q <= 4096: NON-NEGATIVE digits (d0 = h & 127,
  // 4 d1 <= 124 fits an int8) -- a matrix instruction on operands without sign-extension bits costs less energy, and at the power cap
  // that is time: 1.450 against 1.503 ms per 2^20 at N = 821 (1.59 against 1.65 J above idle, tools/power_kernel.py, same device).
  // q = 8192: d1 reaches 63, so h is taken in the representative whose digits fit (d0 in [-64, 63], 4 d1 in [-128, 124]).
  if (q <= 4096) {
    build_toeplitz_array(T0, g, [&](int i) { return (int)(h[i] & (q - 1)) & 127; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
    build_toeplitz_array(T1, g, [&](int i) { return ((int)(h[i] & (q - 1)) >> 7) * 4; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
  } else {
    build_toeplitz_array(T0, g, [&](int i) { const int hs = hs_of(i); return ((hs + 64) & 127) - 64; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
    build_toeplitz_array(T1, g, [&](int i) { const int hs = hs_of(i); const int d0 = ((hs + 64) & 127) - 64; return ((hs - d0) >> 7) * 4; }, (int)threadIdx.x, GROUPS * BLOCK_THREADS);
  }
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
        in_r[j] = load_raw<1>(src_r, pos0 + 16 * lane, 0);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int i = tid * 16 + j * BLOCK_THREADS * 16;
        in_m[j] = load_raw<1>(src_m, src_m.a0 + i, 0);   // past the row block: next rows or zeros, not written
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
        // the instruction count: 8-byte stores of four rows per lane group were 14 % SLOWER (profiles/archive/r02_ablation_*).
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
                {
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
              {
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
              // (profiles/archive/r02_hazard_store_x4_soffset.txt); the compiler only separates the two when soffset is no
              // register (tests/test_build_quality.py scans the ISA for the pattern).
              {
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

#ifdef NTRU_EXPERIMENTS
__global__ __launch_bounds__(BLOCK_THREADS, 2) void k_encrypt_mc(MGeom g, u32 q, const u16 *__restrict__ h,
                                                              const uint8_t *__restrict__ r,
                                                              const uint8_t *__restrict__ m, long B,
                                                              u16 *__restrict__ e, u16 *__restrict__ quotE) {
  encrypt_m_body<4, 1, true>(g, q, h, r, m, B, e, quotE);
}
#endif

__global__ __launch_bounds__(BLOCK_THREADS, 2) void k_encrypt_md(MGeom g, u32 q, const u16 *__restrict__ h,
                                                              const uint8_t *__restrict__ r,
                                                              const uint8_t *__restrict__ m, long B,
                                                              u16 *__restrict__ e, u16 *__restrict__ quotE) {
  encrypt_m_body<4, 1, false, true>(g, q, h, r, m, B, e, quotE);
}

#ifdef NTRU_EXPERIMENTS
__global__ __launch_bounds__(2 * BLOCK_THREADS, 1) void k_encrypt_m8(MGeom g, u32 q, const u16 *__restrict__ h,
                                                                  const uint8_t *__restrict__ r,
                                                                  const uint8_t *__restrict__ m, long B,
                                                                  u16 *__restrict__ e, u16 *__restrict__ quotE) {
  encrypt_m_body<4, 2>(g, q, h, r, m, B, e, quotE);
}
#endif

NTRU_STAMPS_READER(ntru_debug_read_stamps_enc)

#ifdef NTRU_EXPERIMENTS
// ---- family 4, role-split variants ----------------------------------------------------------------------------------
// k_encrypt_m / k_decrypt_m above give every wave the whole job of its column strips: stage, matrix loops, epilogue
// arithmetic, table lookups and 2-byte result stores, phase after phase; the two co-resident workgroups of a CU overlap
// almost none of it (EXPERIMENTS.md, round 1: the ablation times are additive).  The role-split kernels run ONE workgroup
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
    if (drain_rb < 0) return;
    int dk, dn;
    m2_strip(g.NT, 4 * drain_round + w4, &dk, &dn);
    if (want_q) m2_drain_encrypt<true>(chunk, cp, dk, dn, drain_rb << 5, B, g.N, g.ld, q, pre, e, quotE, lane0);
    else m2_drain_encrypt<false>(chunk, cp, dk, dn, drain_rb << 5, B, g.N, g.ld, q, pre, e, quotE, lane0);
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

#endif   // NTRU_EXPERIMENTS

// ---- host side ----------------------------------------------------------------------------------------------------------
// Kernel paths (ntru_engine_set_kernel_path): 0 = auto and 5 -> k_encrypt_md where a row fits one direct-to-LDS instruction, else
// k_encrypt_m; 4 -> k_encrypt_m.  -DNTRU_EXPERIMENTS builds add 6 (k_encrypt_m2, role split), 7 (k_encrypt_mc, results through LDS
// chunks), 9 (k_encrypt_m8, lock-step groups): built, bit-exact, measured slower or equal (EXPERIMENTS.md, round 2).
int ntru_launch_encrypt_matrix(ntru_engine *eng, int N, int q, int ld, const uint16_t *d_h, const uint8_t *d_r, const uint8_t *d_m,
                               int64_t B, uint16_t *d_e, uint16_t *d_quotE) {
  MGeom mg;
  if (!make_mgeom(eng, N, q, ld, &mg)) return NTRU_NOT_TAKEN;
  const size_t lds = (size_t)32 * mg.tpitch + (size_t)32 * mg.pitchA + (((size_t)32 * ld + 15) & ~(size_t)15) + 16;
  const long nrb = (long)((B + 31) / 32);
  dim3 grid;
#ifdef NTRU_EXPERIMENTS
  if (eng->path == 9) {                                  // lock-step variant: two four-wave groups per workgroup, one workgroup per CU
    const size_t per_group = (size_t)32 * mg.pitchA + (((size_t)32 * ld + 15) & ~(size_t)15) + 16;
    const size_t lds8 = (size_t)32 * mg.tpitch + 2 * per_group;
    if (lds8 <= 160 * 1024) {
      if (int rc = resident_grid(eng, k_encrypt_m8, lds8, (nrb + 1) / 2, &grid, 2 * BLOCK_THREADS)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_m8");
      hipLaunchKernelGGL(k_encrypt_m8, grid, dim3(2 * BLOCK_THREADS), lds8, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, d_e, d_quotE);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
  }
  // role-split kernel (one workgroup of 8 waves per CU): needs 16-byte aligned batch arrays and its LDS to fit: 1.9 ms per 2^20
  // at N = 821 against 1.5 ms for k_encrypt_m (profiles/archive/r02_*role_split*).
  if (eng->path == 6 && ((((uintptr_t)d_r | (uintptr_t)d_m | (uintptr_t)d_e | (uintptr_t)d_quotE) & 15) == 0)) {
    size_t lds2 = (size_t)32 * mg.tpitch + (size_t)64 * mg.pitchA;
    for (int w = 0; w < 4; w++) lds2 += (size_t)32 * m2_chunk_pitch(mg.NT, w);
    if (lds2 <= 160 * 1024) {
      if (int rc = resident_grid(eng, k_encrypt_m2, lds2, nrb, &grid, 512)) return rc;
      snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_m2");
      hipLaunchKernelGGL(k_encrypt_m2, grid, dim3(512), lds2, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, d_e, d_quotE);
      HIP_TRY(hipGetLastError());
      return NTRU_OK;
    }
  }
  if (eng->path == 7 && 2 * (lds + WAVES_PER_BLOCK * OC_BYTES) <= 160 * 1024) {      // result chunks: aligned 16-byte stores
    const size_t ldsc = lds + WAVES_PER_BLOCK * OC_BYTES;
    if (int rc = resident_grid(eng, k_encrypt_mc, ldsc, nrb, &grid)) return rc;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_mc");
    hipLaunchKernelGGL(k_encrypt_mc, grid, dim3(BLOCK_THREADS), ldsc, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, d_e, d_quotE);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
#endif
  if (lds > 160 * 1024) return NTRU_NOT_TAKEN;
  // The default: the operands reach LDS by direct-to-LDS loads, r of the next row block ahead of the last epilogue's stores:
  // 1.49-1.51 ms against 1.57-1.62 ms per 2^20 at N = 821 on the same device (profiles/archive/r02_ab_direct_to_lds_rows.txt).
  // One direct-to-LDS instruction moves 64 x 16 bytes from the dword at or below a row, and eight of them per thread the m
  // image: a row of (its byte phase) + N > 1024 bytes, or an image of (phase) + 32 ld > 32768 bytes, would lose its last 1-3
  // bytes.  Those shapes (N >= 1022, or ld = 1024, with rows that are not dword-aligned) take k_encrypt_m, whose register
  // staging fetches the extra dword.
  const bool rows_dword_aligned = (ld & 3) == 0 && ((uintptr_t)d_r & 3) == 0, img_dword_aligned = (ld & 3) == 0 && ((uintptr_t)d_m & 3) == 0;
  const bool dma_fits = (N + 3 <= 1024 || (rows_dword_aligned && N <= 1024)) && (32 * ld + 3 <= 32768 || (img_dword_aligned && 32 * ld <= 32768));
  if (eng->path != 4 && dma_fits) {
    if (int rc = resident_grid(eng, k_encrypt_md, lds, nrb, &grid)) return rc;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_md");
    hipLaunchKernelGGL(k_encrypt_md, grid, dim3(BLOCK_THREADS), lds, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, d_e, d_quotE);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
  if (int rc = resident_grid(eng, k_encrypt_m, lds, nrb, &grid)) return rc;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_encrypt_m");
  hipLaunchKernelGGL(k_encrypt_m, grid, dim3(BLOCK_THREADS), lds, eng->stream, mg, (u32)q, d_h, d_r, d_m, (long)B, d_e, d_quotE);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}
