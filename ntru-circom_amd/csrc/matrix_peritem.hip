// matrix_peritem.hip -- MI355X (gfx950), family 4 with PER-ITEM operands: verifyKeysInputs (index.js:141-197), the generic product
// multiplyPolynomials + dividePolynomials by I (index.js:319-401) and products with a ternary operand (public key, Newton rounds of
// the key inversion), one item per wavefront on the int8 matrix cores.  tools/peritem_mfma_model.py is the executable specification.
#include "matrix_common.h"

// ---- family 4 for PER-ITEM operands: verifyKeysInputs (index.js:141-197) on the matrix cores ---------------------
// No matrix is shared by the batch, but one product c = a * s is itself a 32-row matrix product per tile distance
// d = kb - ib (tools/peritem_mfma_model.py): C[kb][k'] += sum_i' F[kb - d][i'] G_d[i'][k'] with F the 32-coefficient chunks
// of a (rows = output tiles, read as aligned 16-byte pieces of a zero-padded natural-order byte array) and G_d the
// Toeplitz tile of s (fragments of the reversed cyclic array, as above).  One accumulator pair (low / high) holds the whole
// product of an item; a 13-bit operand contributes two digit planes with SEPARATE accumulators (value = acc0 + 128 acc1),
// so nothing is scaled.  2 NT - 1 (+1 for the split diagonal) matrix instructions per plane.  One item per wave, all LDS
// regions private to the wave, no workgroup barrier.
constexpr int PI_PAD = 32;          // zero chunks on either side of the chunk matrix: rows 0..31, distances +-(NT-1)
constexpr int PI_WAVES = 2;         // waves per workgroup (LDS, not registers, bounds the residency: ~15 KB per wave); 4 measured the same
struct PGeom { int N, NT, tpitch; };
static __host__ __device__ inline size_t pi_fa_bytes(const PGeom &g) { return (size_t)32 * (g.NT + 2 * PI_PAD); }
static __host__ __device__ inline size_t pi_nat_bytes(const PGeom &g) { return ((size_t)3 * g.N + 64 + 15) & ~(size_t)15; }
// per wave: the two chunk matrices, then the reversed array.  The three natural-order periods the array is built from
// are staged OVER the chunk matrices (2 fa >= nat for every N <= 1024) and wiped again before the digits go in.
static __host__ __device__ inline size_t pi_wave_bytes(const PGeom &g) { return 2 * pi_fa_bytes(g) + (size_t)16 * g.tpitch; }
// single digit plane (q <= 256): ONE chunk matrix (with room for the natural-order staging that lies over it) and the reversed array
static __host__ __device__ inline size_t pi_one_bytes(const PGeom &g) {
  return (pi_fa_bytes(g) > pi_nat_bytes(g) ? pi_fa_bytes(g) : pi_nat_bytes(g)) + (size_t)16 * g.tpitch;
}

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
#if NTRU_ABLATE & 2                                                       // timing only: no matrix loops
  NT = 1;
#pragma unroll
  for (int i = 0; i < 16; i++) { L1[i] = 0; }
#endif
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

// ---- chunk rows in REGISTERS (k_verify_keys_m) -----------------------------------------------------------------------------------
// The A operand of tile distance d is the chunk matrix moved down by d rows: lane (r, hh) holds bytes 16 hh .. 16 hh + 15 of chunk
// r - d.  Going from d to d + 1 (d >= 0) every lane takes its lower neighbour's 16 bytes and nothing enters at row 0; going from d to
// d - 1 (d <= 0) every lane takes its upper neighbour's and nothing enters at row 31 (chunks >= NT are zero).  So the low part walks
// d = 1, 2, ... and the high part d = -1, -2, ..., each from the unshifted rows, with ONE v_and_b32_dpp per dword and step (wave_shr /
// wave_shl by one lane; the AND cuts the seam between the two half-waves: lane 32 would take lane 31's row, lane 31 lane 32's) and
// no LDS read for the rows at all.  [Round 3 had tried the shift the other way round -- the high part walking d upwards, a new row
// entering at lane 0 every step through a small LDS read and v_cndmask_b32_dpp: slower than reading the rows.]  The loops' LDS
// traffic is the fragment reads alone, shared by every plane that multiplies the same Toeplitz operand: 1 KB per step for the three
// planes of products 1 and 2 (fq lo / hi and fp against f) where the LDS-row form read 5 KB.  bench_micro/peritem_step.hip is the probe.
// A fresh copy of a lane-dependent value that the compiler cannot trace back: everything derived from it is computed where it is used,
// instead of being hoisted out of the item loop (per-lane addresses and masks of every phase: dozens of registers live for ever, i.e.
// spilled at three waves per SIMD).
static __device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

static __device__ __forceinline__ v4i rows_up(v4i a, int seam) {          // lane l <- lane l - 1 (lane 0 <- 0)
  v4i o;
#pragma unroll
  for (int c = 0; c < 4; c++) o[c] = __builtin_amdgcn_update_dpp(0, a[c], 0x138, 0xf, 0xf, true) & seam;
  return o;
}
static __device__ __forceinline__ v4i rows_down(v4i a, int seam) {        // lane l <- lane l + 1 (lane 63 <- 0)
  v4i o;
#pragma unroll
  for (int c = 0; c < 4; c++) o[c] = __builtin_amdgcn_update_dpp(0, a[c], 0x130, 0xf, 0xf, true) & seam;
  return o;
}

// NPL planes F[p] (unshifted chunk rows of this lane, zero at and beyond N) against the Toeplitz fragments of T: L[p] / H[p] = low /
// high half of plane p's product.  Low and high parts advance together (two independent fragment reads and 2 NPL matrix
// instructions per trip, fragments requested one trip ahead, unrolled by two so that the two fragment sets rotate without moves).
template <int NPL>
static __device__ __forceinline__ void pi_product_reg(const v4i (&F)[NPL], const u32 *T, const PGeom &g, int lane_, v16i (&L)[NPL], v16i (&H)[NPL]) {
  const int lane = opaque(lane_), NT = g.NT;
  const int y0 = 32 * NT - 1 - (lane & 31) + 16 * (lane >> 5);
  const u32 *tb = T + (y0 & 3) * g.tpitch + (y0 >> 2);     // this lane's fragment of distance 0; distance d lies 8 d dwords below
  int seam_up = lane == 32 ? 0 : -1, seam_dn = lane == 31 ? 0 : -1;
  asm volatile("" : "+v"(seam_up), "+v"(seam_dn));         // (opaque: as a known 0 / -1 the AND becomes a select that cannot carry the DPP shift)
  auto frag = [&](int d) {                                 // |d| <= NT - 1; requests past the last step read the last fragment again
    d = d > NT - 1 ? NT - 1 : (d < 1 - NT ? 1 - NT : d);
    const u32 *p = tb - 8 * d;
    return (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
  };
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  v4i AL[NPL], AH[NPL];
  const v4i w0 = frag(0);
  v4i wl_a = frag(1), wh_a = frag(-1), wl_b, wh_b;
  {                                                        // d = 0: split by the diagonal mask; the first term of every accumulator
    u32 mlow[4];
    diag_low_mask(lane, mlow);
    const v4i wl = and4(w0, mlow);
    const v4i wh = {(int)((u32)w0[0] & ~mlow[0]), (int)((u32)w0[1] & ~mlow[1]), (int)((u32)w0[2] & ~mlow[2]), (int)((u32)w0[3] & ~mlow[3])};
#pragma unroll
    for (int p = 0; p < NPL; p++) {
      L[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[p], wl, zero, 0, 0, 0);
      H[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[p], wh, zero, 0, 0, 0);
      AL[p] = F[p]; AH[p] = F[p];
    }
  }
  auto trip = [&](const v4i &wl, const v4i &wh) {
#pragma unroll
    for (int p = 0; p < NPL; p++) {
      AL[p] = rows_up(AL[p], seam_up);
      L[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AL[p], wl, L[p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                   // the next plane's shifts issue under this matrix instruction
    }
#pragma unroll
    for (int p = 0; p < NPL; p++) {
      AH[p] = rows_down(AH[p], seam_dn);
      H[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AH[p], wh, H[p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int j = 1;
  for (; j + 1 < NT; j += 2) {
    wl_b = frag(j + 1); wh_b = frag(-(j + 1));
    trip(wl_a, wh_a);
    wl_a = frag(j + 2); wh_a = frag(-(j + 2));
    trip(wl_b, wh_b);
  }
  if (j < NT) trip(wl_a, wh_a);
}

// Per wave: three natural-order periods of the ternary operand (the source of its reversed array; later the remainder of product 3
// for the comparison with h), then the reversed array.  No chunk matrix: the rows live in registers.
static __host__ __device__ inline size_t pi_reg_wave_bytes(const PGeom &g) { return pi_nat_bytes(g) + (size_t)16 * g.tpitch; }

// pi_build_array for a lane that holds chunk ch (any assignment of chunks to lanes); sv zero at and beyond N.  The last chunk of a
// period is stored whole: its zero tail lands on the next period's first bytes, which the NEXT store instruction writes (the LDS
// executes one wave's instructions in order).  nat is not wiped afterwards.
static __device__ __forceinline__ void pi_build_array_ch(unsigned char *nat, u32 *T, const PGeom &g, int lane_, int ch, v4i sv) {
  const int N = g.N, Y0 = 32 * g.NT - 1, lane = opaque(lane_);
  const bool holds = 16 * ch < N;
#pragma unroll
  for (int k = 0; k < 3; k++)
    if (holds) *(v4i *)(nat + k * N + 16 * ch) = sv;
  if (ch < 4) *(v4i *)(nat + 3 * N + 16 * ch) = sv;        // N >= 64
  wave_lds_fence();
  const u32 *D = (const u32 *)nat;
  const int e = __builtin_amdgcn_readfirstlane((Y0 + 2 * N - 3) & 3);
  u32 sel[4]; int dk[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int al = c <= e ? e - c : e - c + 4;
    dk[c] = c <= e ? 0 : -1;
    sel[c] = 0x00010203u + 0x01010101u * (u32)al;
  }
  for (int w = lane; w < g.tpitch; w += 64) {
    int K = (Y0 + 2 * N - 3 - 4 * w) >> 2;
    K = K < 1 ? 1 : K;
    const u32 dm = D[K - 1], d0 = D[K], dp = D[K + 1];
#pragma unroll
    for (int c = 0; c < 4; c++)
      T[c * g.tpitch + w] = dk[c] == 0 ? __builtin_amdgcn_perm(dp, d0, sel[c]) : __builtin_amdgcn_perm(d0, dm, sel[c]);
  }
  wave_lds_fence();
}

// verifyKeysInputs (index.js:141-197) for one key pair per wave.  Lane (r, hh) = 32 hh + r holds chunk ch = 2 r + hh (16 coefficients)
// of every operand row -- the layout the matrix instruction wants its A operand in -- so the rows go from HBM to the matrix cores
// through registers only; the LDS holds the reversed array of the ternary operand (f for products 1 and 2, which share every
// fragment read: three planes, six matrix instructions per trip; then g for product 3).
__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_verify_keys_m(
    PGeom g, u32 q, const int8_t *__restrict__ f, const int8_t *__restrict__ gg, const u16 *__restrict__ fq,
    const uint8_t *__restrict__ fp, const u16 *__restrict__ h, long B, u16 *__restrict__ quot_fq,
    u16 *__restrict__ rem_fq, uint8_t *__restrict__ quot_fp, uint8_t *__restrict__ rem_fp, u16 *__restrict__ quot_h,
    u16 *__restrict__ rem_h, uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *nat = lds + (size_t)wave * pi_reg_wave_bytes(g);
  u32 *T = (u32 *)(nat + pi_nat_bytes(g));
  const int N = g.N, NT = g.NT;
  auto chunk_of = [](int ln) { return 2 * (ln & 31) + (ln >> 5); };      // this lane's chunk of every operand row
  auto index_of = [](int ln) { return 128 * (ln >> 5) + (ln & 31); };    // accumulator register i holds index 32 ((i&3) + 8 (i>>2)) + this
  const long item_step = (long)gridDim.x * PI_WAVES;
  // The first operand rows of an item (fq, f, fp: products 1 and 2) are requested at the end of the PREVIOUS item's last epilogue
  // (a wave that fetched them where it needs them sat idle for a round trip to HBM per item) and stay in these registers across
  // the loop back-edge.
  RawChunks<2> r_fq;
  RawChunks<1> r_f, r_fp;
  auto request_first = [&](long it) {
    const long row = it * N, left = (B - it) * N;
    const int ch = chunk_of(opaque(lane));
    const AlignedSrc s_fq = aligned_src(fq + row, 2 * left), s_f = aligned_src(f + row, left), s_fp = aligned_src(fp + row, left);
    r_fq = load_raw<2>(s_fq, s_fq.a0 + 32 * ch, 0); r_f = load_raw<1>(s_f, s_f.a0 + 16 * ch, 0); r_fp = load_raw<1>(s_fp, s_fp.a0 + 16 * ch, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request_first((long)blockIdx.x * PI_WAVES + wave);
  [[maybe_unused]] int stamp_iter = -1;                    // -DNTRU_STAMPS: phase stamps of the first items (tools/phase_stamps_peritem.py)
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N, left = (B - item) * N;
    u32 fl = 0;
    stamp_iter++;
    STAMP(0);
    auto bytes_of = [&](const RawChunks<1> &rw, const void *p) {
      v4i v[1];
      shift_raw<1>(rw, __builtin_amdgcn_readfirstlane((int)((unsigned long long)p & 15)), v);
      return v[0];
    };
    auto fq_pairs = [&](u32 (&x)[8]) {                     // 16 coefficients per lane as u16 pairs
      v4i v[2];
      shift_raw<2>(r_fq, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(fq + row) & 15)), v);
#pragma unroll
      for (int c = 0; c < 4; c++) { x[c] = (u32)v[0][c]; x[4 + c] = (u32)v[1][c]; }
    };
    // ternary operands: any negative byte is -1 (ValTernary), bytes at and beyond N are zero -- four bytes at a time
    auto ternary = [&](v4i v, const v4i &cmask) {
      v4i o;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const u32 w = (u32)(v[c] & cmask[c]);
        u32 neg = (w >> 7) & 0x01010101u;                  // 1 in every negative byte ...
        neg |= neg << 1; neg |= neg << 2; neg |= neg << 4; // ... spread to 0xFF
        o[c] = (int)(w | neg);
      }
      return o;
    };
    // ---- products 1 and 2: fq * f mod q and fp * f mod p (index.js:158-163): the reversed array of f, planes fq lo / fq hi / fp mod 3
    v4i F[3];
    {
      const int ch = chunk_of(opaque(lane));
      const v4i cmask = col_mask16(16 * ch, N);            // bytes of this lane's chunk that are below N
      pi_build_array_ch(nat, T, g, lane, ch, ternary(bytes_of(r_f, f + row), cmask));
      u32 xq[8];
      fq_pairs(xq);
      pi_digits(xq, q, 1u, 16 * ch, N, F[0], F[1]);
      union { v4i v; unsigned char c[16]; } u; u.v = bytes_of(r_fp, fp + row) & cmask;
#pragma unroll
      for (int j = 0; j < 16; j++) u.c[j] = (unsigned char)((u32)u.c[j] % 3u);
      F[2] = u.v;
    }
    STAMP(1);                                              // operands of products 1 and 2 in place
    v16i L[3], H[3];
    pi_product_reg<3>(F, T, g, lane, L, H);
    STAMP(2);                                              // ... their matrix loops
    // product 3's rows (g; fq again: an L2 hit) have the two epilogues to arrive
    const AlignedSrc s_fq = aligned_src(fq + row, 2 * left), s_g = aligned_src(gg + row, left);
    const int ch3 = chunk_of(opaque(lane));
    r_fq = load_raw<2>(s_fq, s_fq.a0 + 32 * ch3, 0);
    const RawChunks<1> r_g = load_raw<1>(s_g, s_g.a0 + 16 * ch3, 0);
    const int kl = index_of(opaque(lane));
    {
      // stores through one-row descriptors: index k = 32 kb + r is a per-lane offset (128 hh + r) plus a compile-time
      // one per register, indices >= N fall outside the descriptor and are dropped -- no address arithmetic per store
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_fq + row, 2L * N), rs_q = rows_rsrc(quot_fq + row, 2L * N);
      bool nz_hi = false, first_not_one = false;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2)), k = ko + kl;
        const int lo = L[0][i] + 128 * L[1][i], hi = H[0][i] + 128 * H[1][i];
        const u32 rv = (u32)(lo + hi) & (q - 1);
        __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
        nz_hi |= k >= 1 && k < N && rv != 0;
        first_not_one |= k == 0 && rv != 1;
      }
      if (__ballot(nz_hi) != 0 && __ballot(first_not_one) != 0) fl |= NTRU_FLAG_INVALID_FQ;   // length !== 1 && [0] !== 1
    }
    STAMP(3);                                              // product 1's epilogue
    {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_fp + row, (long)N), rs_q = rows_rsrc(quot_fp + row, (long)N);
      bool nz_hi = false, first_not_one = false;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2)), k = ko + kl;
        // |L + H|, |H| <= 127 N (f is an int8, fp < 3): a multiple of 3 above that keeps the dividend non-negative
        const u32 x = (u32)(L[2][i] + H[2][i] + 3 * 131072), y = (u32)(3 * 131072 - H[2][i]);
        const u32 rv = x % 3u, qv = y % 3u;
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)rv, rs_r, kl, ko, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)qv, rs_q, kl, ko, 0);
        nz_hi |= k >= 1 && k < N && rv != 0;
        first_not_one |= k == 0 && rv != 1;
      }
      if (__ballot(nz_hi) != 0 && __ballot(first_not_one) != 0) fl |= NTRU_FLAG_INVALID_FP;
    }
    STAMP(4);                                              // product 2's epilogue
    // ---- product 3: ((p fq) mod q) * g mod q, compared with h below its trimmed length (index.js:155,164-166)
    v4i G[2];
    {
      const int ch = chunk_of(opaque(lane));
      pi_build_array_ch(nat, T, g, lane, ch, ternary(bytes_of(r_g, gg + row), col_mask16(16 * ch, N)));
      u32 xq[8];
      fq_pairs(xq);
      pi_digits(xq, q, 3u, 16 * ch, N, G[0], G[1]);
    }
    // h is requested before the product whose remainder it is compared with, as a natural-order row chunk (16 coefficients per lane);
    // the remainder gets into the same layout through the wave's LDS (the natural-order area is free again by then)
    const AlignedSrc s_h = aligned_src(h + row, 2 * left);
    const RawChunks<2> r_h = load_raw<2>(s_h, s_h.a0 + 32 * opaque(lane), 0);
    STAMP(5);                                              // product 3: operands in place
    v16i L3[2], H3[2];
    pi_product_reg<2>(G, T, g, lane, L3, H3);
    STAMP(6);
    {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_h + row, 2L * N), rs_q = rows_rsrc(quot_h + row, 2L * N);
      u16 *remx = (u16 *)nat;
      const int kl = index_of(opaque(lane));
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const int lo = L3[0][i] + 128 * L3[1][i], hi = H3[0][i] + 128 * H3[1][i];
        const u32 rv = (u32)(lo + hi) & (q - 1);
        __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
        remx[ko + kl] = (u16)rv;                           // ko + kl < 32 NT <= (3 N + 64) / 2
      }
      if (item + item_step < B) request_first(item + item_step);   // the next item's first rows: in flight from here on (the accumulators are dead)
      wave_lds_fence();
      // index.js:165: h[k] must equal the remainder for every k below h's trimmed length
      v4i hc[2];
      shift_raw<2>(r_h, __builtin_amdgcn_readfirstlane(s_h.a0), hc);
      const int i0 = 16 * opaque(lane);
      u32 nz = 0, df = 0;                                  // bit j: h[i0 + j] != 0 / != remainder[i0 + j]
      if (i0 < 32 * NT) {
        const v4i rc0 = *(const v4i *)(nat + 2 * i0), rc1 = *(const v4i *)(nat + 2 * i0 + 16);
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
      const int hl = wtop >= 0 ? wtop + 1 : 1;             // trimmed length of h (1 for the zero polynomial)
      const int nv = hl - i0 < 0 ? 0 : (hl - i0 > 16 ? 16 : hl - i0);   // this lane's indices below hl
      if (__ballot((df & ((1u << nv) - 1u)) != 0) != 0) fl |= NTRU_FLAG_INVALID_H;
    }
    if (lane == 0) flags[item] = (uint8_t)fl;
    wave_lds_fence();
    STAMP(7);                                              // product 3's epilogue and the comparison with h
  }
}

// One per-item product on the matrix cores: rem (and quot) of ((mul a) mod q) * s split by 1 - x^N, a < 2^16 per item,
// s ternary per item: generatePublicKeyH (index.js:72-79, mul = p) and the f * t product of polyInv's Newton rounds
// (index.js:499-506, mul = 1; there the remainder leaves as (f v - 1) / 2^nshift, see ntru_invert_key_batch_dev).  Same machinery
// as k_verify_keys_m.
// ONE: a single int8 digit plane (q <= 256: the early Newton rounds): half the accumulators, no second chunk matrix -- 128 registers and
// 9.7 KB of LDS per wave, i.e. FOUR waves per SIMD instead of three.
template <bool ONE>
__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(ONE ? 4 : 3, 4))) void k_product_tern_m(
    PGeom g, u32 q, u32 mul, u32 nshift, const u16 *__restrict__ a, const int8_t *__restrict__ s, long B, u16 *__restrict__ quot,
    u16 *__restrict__ rem) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
  const size_t fa_region = ONE ? pi_one_bytes(g) - (size_t)16 * g.tpitch : 2 * pi_fa_bytes(g);
  unsigned char *fa0 = lds + (size_t)wave * (ONE ? pi_one_bytes(g) : pi_wave_bytes(g)), *fa1 = ONE ? fa0 : fa0 + pi_fa_bytes(g), *nat = fa0;
  u32 *T = (u32 *)(fa0 + fa_region);
  const int N = g.N, NT = g.NT;
  for (size_t i = 16 * lane; i < fa_region; i += 16 * 64) *(v4i *)(fa0 + i) = (v4i){0, 0, 0, 0};   // the pads stay zero
  const int y0 = 32 * NT - 1 - r + 16 * hh;
  const u32 *tb = T + (y0 & 3) * g.tpitch + (y0 >> 2);
  const unsigned char *pa0 = fa0 + 32 * PI_PAD + 32 * r + 16 * hh, *pa1 = fa1 + 32 * PI_PAD + 32 * r + 16 * hh;
  u32 mlow[4];
  diag_low_mask(lane, mlow);
  const bool stager = 16 * lane < 32 * NT;
  const v4i cmask = col_mask16(16 * lane, N);
  const bool want_q = quot != nullptr;
  wave_lds_fence();
  // The operands of the NEXT item are requested as soon as this item's are in LDS (the round trip to HBM runs under the matrix
  // loops and the result stores instead of in front of every item).
  const long item_step = (long)gridDim.x * PI_WAVES;
  RawChunks<2> ra;
  RawChunks<1> rs;
  auto request = [&](long it) {
    const long rw = it * N, lf = (B - it) * N;
    const AlignedSrc sa = aligned_src(a + rw, 2 * lf), ss = aligned_src(s + rw, lf);
    ra = load_raw<2>(sa, sa.a0 + 32 * lane, 0);
    rs = load_raw<1>(ss, ss.a0 + 16 * lane, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request((long)blockIdx.x * PI_WAVES + wave);
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N;
    {
      v4i va[2], vs[1];
      shift_raw<2>(ra, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(a + row) & 15)), va);
      shift_raw<1>(rs, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(s + row) & 15)), vs);
      if (item + item_step < B) request(item + item_step);
      u32 xa[8];
#pragma unroll
      for (int c = 0; c < 4; c++) { xa[c] = (u32)va[0][c]; xa[4 + c] = (u32)va[1][c]; }
      union { v4i v; signed char c[16]; } u; u.v = vs[0] & cmask;        // any negative byte is -1 (ValTernary)
#pragma unroll
      for (int j = 0; j < 16; j++) u.c[j] = u.c[j] < 0 ? (signed char)-1 : u.c[j];
      pi_build_array(nat, T, g, lane, u.v);
      if (stager) {
        v4i o0, o1;
        // (ONE with q > 256: the caller vouches for a < 128 -- a Newton round whose v has at most 7 bits -- and such an a is its own
        // centred representative modulo 256; the result is still reduced modulo q)
        pi_digits(xa, ONE && q > 256 ? 256u : q, mul, 16 * lane, N, o0, o1);
        *(v4i *)(fa0 + 32 * PI_PAD + 16 * lane) = o0;
        if (!ONE) *(v4i *)(fa1 + 32 * PI_PAD + 16 * lane) = o1;
      }
      wave_lds_fence();
    }
    v16i L0, L1, H0, H1;
    pi_product<!ONE>(pa0, pa1, tb, NT, mlow, L0, L1, H0, H1);       // ONE: one digit plane (early Newton rounds, small q)
    {
      const int kl = 128 * hh + r;                                       // see k_verify_keys_m: indices >= N are dropped
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem + row, 2L * N);
      const __amdgpu_buffer_rsrc_t rs_q = rows_rsrc(want_q ? quot + row : nullptr, want_q ? 2L * N : 0L);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const int lo = L0[i] + 128 * L1[i], hi = H0[i] + 128 * H1[i];
        u32 rv = (u32)(lo + hi) & (q - 1);
        // nshift = k > 0 (a Newton round of the key inversion, a = v with f v = 1 mod 2^k): what is stored is e = (f v - 1) / 2^k
        if (nshift) rv = ((rv - (i == 0 && kl == 0 ? 1u : 0u)) & (q - 1)) >> nshift;
        __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
        if (want_q) __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);   // (a store through an empty descriptor is dropped, but issued)
      }
    }
    wave_lds_fence();
  }
}

// One Newton round of the key inversion (polyInv, index.js:499-506) per item in ONE kernel, in its lifted form: v is right modulo
// 2^kb (kb <= 7 bits, the schedule of ntru_invert_key_batch_dev), f v = 1 + 2^kb e, and v <- v - 2^kb (e v mod 2^(m - kb)) is right
// modulo 2^m, m <= 2 kb.  Both products run on one int8 digit plane: f (x) v with f's reversed array and v (< 128) as the chunk matrix;
// then e -- at most kb bits per coefficient -- goes from the accumulator layout STRAIGHT into the chunk matrix as bytes (no
// natural-order pass: the chunk matrix IS natural order), v's residues modulo 2^(m - kb) become the reversed array in f's place, and
// the second product's epilogue lifts v on the item's own row.  Against the two kernels it replaces: v and f are fetched once, e never
// leaves the CU, one launch, one set of per-item waits.
__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_newton_round_m(
    PGeom g, u32 kb, u32 m, const int8_t *__restrict__ f, u16 *v, long B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
  const size_t fa_region = pi_one_bytes(g) - (size_t)16 * g.tpitch;
  unsigned char *fa0 = lds + (size_t)wave * pi_one_bytes(g), *nat = fa0;
  u32 *T = (u32 *)(fa0 + fa_region);
  const int N = g.N, NT = g.NT;
  const u32 mr = 1u << m, me = 1u << (m - kb);
  for (size_t i = 16 * lane; i < fa_region; i += 16 * 64) *(v4i *)(fa0 + i) = (v4i){0, 0, 0, 0};   // the pads stay zero
  const int y0 = 32 * NT - 1 - r + 16 * hh;
  const u32 *tb = T + (y0 & 3) * g.tpitch + (y0 >> 2);
  const unsigned char *pa0 = fa0 + 32 * PI_PAD + 32 * r + 16 * hh;
  u32 mlow[4];
  diag_low_mask(lane, mlow);
  const bool stager = 16 * lane < 32 * NT;
  const int kl = 128 * hh + r;                             // accumulator register i holds index 32 ((i&3) + 8 (i>>2)) + kl
  wave_lds_fence();
  const long item_step = (long)gridDim.x * PI_WAVES;
  RawChunks<2> rv;
  RawChunks<1> rf;
  auto request = [&](long it) {
    const long rw = it * N, lf = (B - it) * N;
    const AlignedSrc sv = aligned_src(v + rw, 2 * lf), sf = aligned_src(f + rw, lf);
    rv = load_raw<2>(sv, sv.a0 + 32 * lane, 0);
    rf = load_raw<1>(sf, sf.a0 + 16 * lane, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request((long)blockIdx.x * PI_WAVES + wave);
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N;
    v4i b0;                                                // v modulo 2^(m - kb), centred: the second product's Toeplitz operand
    int ln = lane;                                         // (opaque per item: the column masks of pi_digits are otherwise hoisted out of the
    asm volatile("" : "+v"(ln));                           //  item loop -- 24 registers live for ever, spilled at 128 per wave)
    {
      v4i va[2], vf[1];
      shift_raw<2>(rv, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(v + row) & 15)), va);
      shift_raw<1>(rf, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(f + row) & 15)), vf);
      u32 xv[8];
#pragma unroll
      for (int c = 0; c < 4; c++) { xv[c] = (u32)va[0][c]; xv[4 + c] = (u32)va[1][c]; }
      {
        v4i b1;
        pi_digits(xv, me, 1u, 16 * ln, N, b0, b1);
      }
      union { v4i v; signed char c[16]; } u; u.v = vf[0] & col_mask16(16 * ln, N);        // any negative byte is -1 (ValTernary)
#pragma unroll
      for (int j = 0; j < 16; j++) u.c[j] = u.c[j] < 0 ? (signed char)-1 : u.c[j];
      pi_build_array(nat, T, g, lane, u.v);
      if (stager) {
        v4i o0, o1;
        pi_digits(xv, 256u, 1u, 16 * ln, N, o0, o1);     // v < 2^kb <= 128: its own centred representative modulo 256
        *(v4i *)(fa0 + 32 * PI_PAD + 16 * lane) = o0;
      }
      wave_lds_fence();
    }
    v16i L0, L1, H0, H1;
    pi_product<false>(pa0, pa0, tb, NT, mlow, L0, L1, H0, H1);           // f v
    if (item + item_step < B) request(item + item_step);   // the next item's rows (nobody lifts them before this wave does): a product ahead
    u32 e[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const u32 fv = (u32)(L0[i] + H0[i]) & (mr - 1);
      e[i] = ((fv - (i == 0 && kl == 0 ? 1u : 0u)) & (mr - 1)) >> kb;   // e = (f v - 1) / 2^kb, below 2^(m - kb) <= 128
    }
    {
      pi_build_array(nat, T, g, lane, b0);                 // (over f's array; it leaves the chunk matrix region zero)
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int k = 32 * ((i & 3) + 8 * (i >> 2)) + kl;
        if (k < N) fa0[32 * PI_PAD + k] = (unsigned char)e[i];
      }
      wave_lds_fence();
    }
    // v in the accumulator layout, for the lift (the row this item staged a moment ago: an L2 hit), in flight during the second product
    const __amdgpu_buffer_rsrc_t rs_v = rows_rsrc(v + row, 2L * N);
    u16 vold[16];
#pragma unroll
    for (int i = 0; i < 16; i++) vold[i] = (u16)__builtin_amdgcn_raw_buffer_load_b16(rs_v, 2 * kl, 2 * 32 * ((i & 3) + 8 * (i >> 2)), 0);
    pi_product<false>(pa0, pa0, tb, NT, mlow, L0, L1, H0, H1);           // e v
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int ko = 32 * ((i & 3) + 8 * (i >> 2));
      const u32 w = (u32)(L0[i] + H0[i]) & (me - 1);
      __builtin_amdgcn_raw_buffer_store_b16((u16)(((u32)vold[i] - (w << kb)) & (mr - 1)), rs_v, 2 * kl, 2 * ko, 0);
    }
    wave_lds_fence();
  }
}

// Generic per-item product on the matrix cores: both operands < q <= 8192 (multiplyPolynomials + dividePolynomials by I,
// index.js:319-401, with q a power of two; the v * v product of polyInv's Newton rounds).  With a = a0 + 128 a1 and
// b = b0 + 128 b1 the product is a0 b0 + 128 (a0 b1 + a1 b0) + 16384 a1 b1, and 16384 = 0 mod q: three plane products, two
// accumulator groups, two reversed arrays (the digit planes of b) per item.
static __host__ __device__ inline size_t pi_wave_bytes2(const PGeom &g) { return pi_wave_bytes(g) + (size_t)16 * g.tpitch; }

// ONE (q <= 256: one digit plane per operand -- every Newton round of the key inversion in its lifted form): no second chunk matrix,
// no second reversed array, half the accumulators: 9.7 instead of 19.3 KB of LDS per wave and 128 registers, i.e. sixteen instead of
// eight waves per CU.
template <bool ONE>
__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(ONE ? 4 : 3, 4))) void k_polymul_m(
    PGeom g, u32 q, const u16 *__restrict__ a, const u16 *b, long B, u16 *__restrict__ quot,
    u16 *__restrict__ rem, u16 *lift_v, u32 lift_k, u32 lift_q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;
  const size_t fa_region = ONE ? pi_one_bytes(g) - (size_t)16 * g.tpitch : 2 * pi_fa_bytes(g);
  unsigned char *fa0 = lds + (size_t)wave * (ONE ? pi_one_bytes(g) : pi_wave_bytes2(g)), *fa1 = ONE ? fa0 : fa0 + pi_fa_bytes(g), *nat = fa0;
  u32 *T0 = (u32 *)(fa0 + fa_region), *T1 = T0 + 4 * g.tpitch;
  const int N = g.N, NT = g.NT;
  for (size_t i = 16 * lane; i < fa_region; i += 16 * 64) *(v4i *)(fa0 + i) = (v4i){0, 0, 0, 0};   // the pads stay zero
  const int y0 = 32 * NT - 1 - r + 16 * hh;
  const u32 *tb0 = T0 + (y0 & 3) * g.tpitch + (y0 >> 2), *tb1 = tb0 + 4 * g.tpitch;
  const unsigned char *pa0 = fa0 + 32 * PI_PAD + 32 * r + 16 * hh, *pa1 = fa1 + 32 * PI_PAD + 32 * r + 16 * hh;
  u32 mlow[4];
  diag_low_mask(lane, mlow);
  const bool stager = 16 * lane < 32 * NT;
  constexpr bool one = ONE;                                // single int8 plane per operand (pi_digits)
  const bool want_q = quot != nullptr;                     // the Newton rounds of the key inversion only need the remainder
  wave_lds_fence();
  const long item_step = (long)gridDim.x * PI_WAVES;         // the NEXT item's operands are requested early: see k_product_tern_m
  RawChunks<2> rwa, rwb;
  auto request = [&](long it) {
    const long rw = it * N, lf = (B - it) * N;
    const AlignedSrc sa = aligned_src(a + rw, 2 * lf), sb = aligned_src(b + rw, 2 * lf);
    rwa = load_raw<2>(sa, sa.a0 + 32 * lane, 0);
    rwb = load_raw<2>(sb, sb.a0 + 32 * lane, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request((long)blockIdx.x * PI_WAVES + wave);
  [[maybe_unused]] int stamp_iter = -1;                    // -DNTRU_STAMPS: phase stamps of the first items (tools/phase_stamps_peritem.py)
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N;
    stamp_iter++;
    STAMP(0);
    {
      u32 xa[8], xb[8];
      {
        v4i va[2], vb[2];
        shift_raw<2>(rwa, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(a + row) & 15)), va);
        shift_raw<2>(rwb, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(b + row) & 15)), vb);
#pragma unroll
        for (int c = 0; c < 4; c++) { xa[c] = (u32)va[0][c]; xa[4 + c] = (u32)va[1][c]; xb[c] = (u32)vb[0][c]; xb[4 + c] = (u32)vb[1][c]; }
      }
      STAMP(1);                                            // operands arrived
      if (item + item_step < B) request(item + item_step);
      v4i a0, a1, b0, b1;
      pi_digits(xa, q, 1u, 16 * lane, N, a0, a1);
      pi_digits(xb, q, 1u, 16 * lane, N, b0, b1);
      STAMP(2);                                            // digits
      pi_build_array(nat, T0, g, lane, b0);
      if (!one) pi_build_array(nat, T1, g, lane, b1);
      STAMP(3);                                            // reversed arrays
      if (stager) {
        *(v4i *)(fa0 + 32 * PI_PAD + 16 * lane) = a0;
        if (!one) *(v4i *)(fa1 + 32 * PI_PAD + 16 * lane) = a1;
      }
      wave_lds_fence();
    }
    STAMP(4);                                              // chunk matrices
    v16i L0, L1, H0, H1;                                   // group 0: a0 b0; group 1: a0 b1 + a1 b0
#pragma unroll
    for (int i = 0; i < 16; i++) { L0[i] = 0; L1[i] = 0; H0[i] = 0; H1[i] = 0; }
    // The loop lives on its LDS reads (phase stamps, profiles/r03_phase_stamps_peritem.txt: ~250 clocks per step = what 12 waves x 4 KB
    // cost the CU's LDS), so the single-plane form (q <= 256: the early Newton rounds) must not read the second plane's operands.
    auto loops = [&]() {
      auto ld = [&](int d, v4i &x0, v4i &x1, v4i &w0, v4i &w1) {
        const u32 *p0 = tb0 - 8 * d;
        w0 = (v4i){(int)p0[0], (int)p0[1], (int)p0[2], (int)p0[3]};
        x0 = *(const v4i *)(pa0 - 32 * d);
        if (!ONE) {
          const u32 *p1 = tb1 - 8 * d;
          w1 = (v4i){(int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
          x1 = *(const v4i *)(pa1 - 32 * d);
        }
      };
      auto mm3 = [&](v16i &X0, v16i &X1, v4i x0, v4i x1, v4i w0, v4i w1) {
        X0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x0, w0, X0, 0, 0, 0);
        if (!ONE) {                                        // q <= 256: both operands are single planes
          X1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x0, w1, X1, 0, 0, 0);
          X1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x1, w0, X1, 0, 0, 0);
        }
      };
      v4i x0, x1 = {0, 0, 0, 0}, w0, w1 = {0, 0, 0, 0};
      ld(-(NT - 1), x0, x1, w0, w1);
      for (int d = -(NT - 1); d < 0; d++) {
        v4i n0, n1 = {0, 0, 0, 0}, m0, m1 = {0, 0, 0, 0};
        ld(d + 1, n0, n1, m0, m1);
        mm3(H0, H1, x0, x1, w0, w1);
        x0 = n0; x1 = n1; w0 = m0; w1 = m1;
      }
      {
        v4i n0, n1 = {0, 0, 0, 0}, m0, m1 = {0, 0, 0, 0};
        ld(1, n0, n1, m0, m1);
        u32 mhigh[4];
#pragma unroll
        for (int c = 0; c < 4; c++) mhigh[c] = ~mlow[c];
        mm3(L0, L1, x0, x1, and4(w0, mlow), and4(w1, mlow));
        mm3(H0, H1, x0, x1, and4(w0, mhigh), and4(w1, mhigh));
        x0 = n0; x1 = n1; w0 = m0; w1 = m1;
      }
      for (int d = 1; d < NT; d++) {
        v4i n0, n1 = {0, 0, 0, 0}, m0, m1 = {0, 0, 0, 0};
        ld(d + 1, n0, n1, m0, m1);
        mm3(L0, L1, x0, x1, w0, w1);
        x0 = n0; x1 = n1; w0 = m0; w1 = m1;
      }
    };
    // lift_v (a Newton round of the key inversion in its lifted form, ntru_invert_key_batch_dev): what leaves is not the product w but
    // v <- (v - 2^lift_k w) mod lift_q on the item's row of lift_v, read here in the accumulator layout (it is the b row this item
    // staged a moment ago: an L2 hit) -- the round's third kernel and its two passes over HBM are gone.
    const int kl = 128 * hh + r;                                         // see k_verify_keys_m: indices >= N are dropped
    u16 vold[16];
    if (lift_v) {
      const __amdgpu_buffer_rsrc_t rs_v = rows_rsrc(lift_v + row, 2L * N);
#pragma unroll
      for (int i = 0; i < 16; i++) vold[i] = (u16)__builtin_amdgcn_raw_buffer_load_b16(rs_v, 2 * kl, 2 * 32 * ((i & 3) + 8 * (i >> 2)), 0);
    }
    loops();
    STAMP(5);                                              // matrix loops
    if (lift_v) {
      const __amdgpu_buffer_rsrc_t rs_v = rows_rsrc(lift_v + row, 2L * N);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const u32 w = ((u32)L0[i] + 128u * (u32)L1[i] + (u32)H0[i] + 128u * (u32)H1[i]) & (q - 1);
        __builtin_amdgcn_raw_buffer_store_b16((u16)(((u32)vold[i] - (w << lift_k)) & (lift_q - 1)), rs_v, 2 * kl, 2 * ko, 0);
      }
    } else {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem + row, 2L * N);
      const __amdgpu_buffer_rsrc_t rs_q = rows_rsrc(want_q ? quot + row : nullptr, want_q ? 2L * N : 0L);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const u32 lo = (u32)L0[i] + 128u * (u32)L1[i], hi = (u32)H0[i] + 128u * (u32)H1[i];
        __builtin_amdgcn_raw_buffer_store_b16((u16)((lo + hi) & (q - 1)), rs_r, 2 * kl, 2 * ko, 0);
        if (want_q) __builtin_amdgcn_raw_buffer_store_b16((u16)((0u - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
      }
    }
    wave_lds_fence();
    STAMP(6);                                              // result stores issued
  }
}
NTRU_STAMPS_READER(ntru_debug_read_stamps_pi)

// ---- host side ----------------------------------------------------------------------------------------------------------
static PGeom make_pgeom(int N) {
  PGeom pg;
  pg.N = N; pg.NT = (N + 31) / 32; pg.tpitch = ((16 * pg.NT + 31) / 32) * 32 + 8;
  return pg;
}
// The per-item matrix kernels: modulus a power of two <= 8192 (two int8 digit planes), 64 <= N <= 1024; automatic from N = 128.
static bool peritem_applies(const ntru_engine *eng, int N, int q) {
  return (eng->path == 0 || eng->path >= 4) && is_pow2(q) && q <= 8192 && N <= 1024 && N >= (eng->path >= 4 ? 64 : 128);
}
template <class Kern>
static int peritem_grid(ntru_engine *eng, Kern kern, size_t lds, long B, dim3 *grid) {
  int per_cu = 0;
  if (int rc = ntru_blocks_per_cu(eng, (const void *)kern, 64 * PI_WAVES, lds, &per_cu)) return rc;
  long blocks = (long)eng->cus * (per_cu < 1 ? 1 : per_cu), work = (B + PI_WAVES - 1) / PI_WAVES;
  if (blocks > work) blocks = work;
  *grid = dim3((unsigned)blocks);
  return NTRU_OK;
}

int ntru_launch_polymul_matrix(ntru_engine *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b, int64_t B, uint16_t *d_quot,
                               uint16_t *d_rem, uint16_t *d_lift_v, int lift_k, int lift_q) {
  if (!peritem_applies(eng, N, mod)) return NTRU_NOT_TAKEN;
  const PGeom pg = make_pgeom(N);
  dim3 grid;
  auto go = [&](auto kern, size_t lds) -> int {
    if (int rc = peritem_grid(eng, kern, lds, (long)B, &grid)) return rc;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_polymul_m");
    hipLaunchKernelGGL(kern, grid, dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)mod, d_a, d_b, (long)B, d_quot, d_rem, (u16 *)d_lift_v,
                       (u32)lift_k, (u32)lift_q);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  };
  return mod <= 256 ? go(k_polymul_m<true>, PI_WAVES * pi_one_bytes(pg)) : go(k_polymul_m<false>, PI_WAVES * pi_wave_bytes2(pg));
}

bool ntru_product_tern_matrix_applies(const ntru_engine *eng, int N, int q) { return peritem_applies(eng, N, q); }

int ntru_launch_product_tern_matrix(ntru_engine *eng, int N, int q, uint32_t mul, const uint16_t *d_a, const int8_t *d_s, long B,
                                    uint16_t *d_quot, uint16_t *d_rem, uint32_t nshift, int abits) {
  const PGeom pg = make_pgeom(N);
  dim3 grid;
  auto go = [&](auto kern, size_t lds) -> int {
    if (int rc = peritem_grid(eng, kern, lds, B, &grid)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)q, (u32)mul, (u32)nshift, d_a, d_s, B, d_quot, d_rem);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  };
  // one digit plane: q <= 256, or every a[i] below 2^abits <= 128 (then mul must be 1)
  return q <= 256 || (abits <= 7 && mul == 1u) ? go(k_product_tern_m<true>, PI_WAVES * pi_one_bytes(pg))
                                               : go(k_product_tern_m<false>, PI_WAVES * pi_wave_bytes(pg));
}

int ntru_launch_verify_keys_matrix(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const int8_t *d_g, const uint16_t *d_fq,
                                   const uint8_t *d_fp, const uint16_t *d_h, int64_t B, uint16_t *d_quot_fq, uint16_t *d_rem_fq,
                                   uint8_t *d_quot_fp, uint8_t *d_rem_fp, uint16_t *d_quot_h, uint16_t *d_rem_h, uint8_t *d_flags) {
  if (p != 3 || !peritem_applies(eng, N, q)) return NTRU_NOT_TAKEN;
  const PGeom pg = make_pgeom(N);
  const size_t lds = PI_WAVES * pi_reg_wave_bytes(pg);
  dim3 grid;
  if (int rc = peritem_grid(eng, k_verify_keys_m, lds, (long)B, &grid)) return rc;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_verify_keys_m");
  hipLaunchKernelGGL(k_verify_keys_m, grid, dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)q, d_f, d_g, d_fq, d_fp, d_h, (long)B,
                     d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp, d_quot_h, d_rem_h, d_flags);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}


// One Newton round (v from kb to m bits) of the key inversion as ONE kernel: kb <= 7 (v below 128: one digit plane), per-item matrix path.
int ntru_launch_newton_round_matrix(ntru_engine *eng, int N, int kb, int m, const int8_t *d_f, uint16_t *d_v, long B) {
  if (kb > 7 || m > 2 * kb || m <= kb || !peritem_applies(eng, N, 1 << m)) return NTRU_NOT_TAKEN;
  const PGeom pg = make_pgeom(N);
  const size_t lds = PI_WAVES * pi_one_bytes(pg);
  dim3 grid;
  if (int rc = peritem_grid(eng, k_newton_round_m, lds, B, &grid)) return rc;
  hipLaunchKernelGGL(k_newton_round_m, grid, dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)kb, (u32)m, d_f, d_v, B);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}
