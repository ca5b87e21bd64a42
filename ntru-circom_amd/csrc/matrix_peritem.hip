// matrix_peritem.hip -- MI355X (gfx950), family 4 with PER-ITEM operands: verifyKeysInputs (index.js:141-197), the generic product
// multiplyPolynomials + dividePolynomials by I (index.js:319-401) and products with a ternary operand (public key, Newton rounds of
// the key inversion), one item per wavefront on the int8 matrix cores.  tools/peritem_mfma_model.py is the executable specification.
#include "matrix_common.h"

// ---- family 4 for PER-ITEM operands: verifyKeysInputs (index.js:141-197) on the matrix cores ---------------------
// No matrix is shared by the batch, but one product c = a * s is itself a 32-row matrix product per tile distance
// d = kb - ib (tools/peritem_mfma_model.py): C[kb][k'] += sum_i' F[kb - d][i'] G_d[i'][k'] with F the 32-coefficient chunks
// of a (rows = output tiles) and G_d the Toeplitz tile of s (fragments of the reversed cyclic array the wave builds per item in
// its own LDS, as in the shared-key kernels).  One accumulator pair (low / high) holds the whole product of an item; a 13-bit
// operand contributes two digit planes with SEPARATE accumulators (value = acc0 + 128 acc1), so nothing is scaled.
// 2 NT - 1 (+1 for the split diagonal) matrix instructions per plane.  One item per wave, all LDS regions private to the wave,
// no workgroup barrier.  The chunk rows never touch the LDS: see "chunk rows in REGISTERS" below.
constexpr int PI_WAVES = 2;         // waves per workgroup (registers bound the residency); 4 measured the same
struct PGeom { int N, NT, tpitch; };
// Natural-order area of a wave: three periods + 64 bytes of the ternary / Toeplitz operand while its reversed array is built; later a
// product's results as a natural-order image written from the accumulator layout, where register i of lane (r, hh) holds index
// 32 ((i&3) + 8 (i>>2)) + 128 hh + r <= 1151 WHATEVER N is (tiles at and beyond NT hold junk that nobody reads): 2304 bytes of u16.
static __host__ __device__ inline size_t pi_nat_bytes(const PGeom &g) {
  const size_t periods = ((size_t)3 * g.N + 64 + 15) & ~(size_t)15;
  return periods > 2304 ? periods : 2304;
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

// Ternary operand bytes: any negative byte is -1 (ValTernary), bytes outside the mask (at and beyond N) are zero -- four at a time.
static __device__ __forceinline__ v4i pi_ternary(v4i v, const v4i &cmask) {
  v4i o;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const u32 w = (u32)(v[c] & cmask[c]);
    o[c] = (int)(w | ((w >> 7) & 0x01010101u) * 0xFFu);    // (1 in every negative byte, spread to 0xFF: no carries)
  }
  return o;
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

// diag_low_mask (matrix_common.h) for kernels that make the mask once per PRODUCT instead of once per launch: byte jj of dword c is
// set iff r >= 16 hh + 4 c + jj, i.e. the n = clamp(r - 16 hh - 4 c + 1, 0, 4) low bytes -- a handful of instructions per dword
// where sixteen byte-wise compare / select pairs cost ~50.
static __device__ __forceinline__ void pi_diag_low_mask(int lane, u32 (&mlow)[4]) {
  const int u = (lane & 31) - 16 * (lane >> 5) + 1;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int n = min(max(u - 4 * c, 0), 4);
    mlow[c] = n >= 4 ? 0xFFFFFFFFu : ((1u << (8 * n)) - 1u);
  }
}

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
    pi_diag_low_mask(lane, mlow);
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

// The half of the reversed array that the distances d >= 0 read (products modulo x^N - 1: pi_product_cyc): words w < 8 NT + 8 of
// every copy, which look at stream bytes 2N - 36 .. 3N + 34 only -- ONE period and two margins.  So the period is stored once, at the
// 16-byte aligned PI_HALF_BASE, and the margins (the last <= 4 chunks in front of it, the first three behind it) by ONE store
// instruction of other lanes: an LDS access of 16 bytes off its natural alignment is replayed at 64 cycles per wave instruction
// (MI355X guide, LDS), and the three-period form pays that three times per array.  With the period aligned the byte phase of the
// words is 0 for every N: constant selectors.  (The margins first: the zero tail of the last chunk lands on the period's first
// bytes, which the period's own store then writes.)
constexpr int PI_HALF_BASE = 64;
static __device__ __forceinline__ void pi_build_array_half(unsigned char *nat, u32 *T, const PGeom &g, int lane_, int ch, v4i sv) {
  const int N = g.N, lane = opaque(lane_);
  const bool holds = 16 * ch < N, behind = ch < 3, front = holds && 16 * ch + 15 >= N - 36;
  if (N >= 160) {                                          // (chunks 0 .. 2 are none of the last four)
    if (behind || front) *(v4i *)(nat + PI_HALF_BASE + (behind ? N : -N) + 16 * ch) = sv;
  } else {
    if (front) *(v4i *)(nat + PI_HALF_BASE - N + 16 * ch) = sv;
    if (behind) *(v4i *)(nat + PI_HALF_BASE + N + 16 * ch) = sv;
  }
  if (holds) *(v4i *)(nat + PI_HALF_BASE + 16 * ch) = sv;
  wave_lds_fence();
  // Word w of copy c holds stream bytes A .. A + 3 reversed, A = E - c, E = PI_HALF_BASE + 32 NT - 4 - 4 w: a multiple of 4.
  const u32 *D = (const u32 *)nat;
  const int K0 = (PI_HALF_BASE >> 2) + 8 * g.NT - 1;
  for (int w0 = 4 * lane; w0 < 8 * g.NT + 8; w0 += 256) {
    const int base = K0 - w0 - 4;                          // >= 3; words w0 + j need dwords K0 - w0 - j - 1, K0 - w0 - j
    u32 d[6];
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = D[base + i];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      v4i o;
#pragma unroll
      for (int j = 0; j < 4; j++)
        o[j] = (int)(c == 0 ? __builtin_amdgcn_perm(d[5 - j], d[4 - j], 0x00010203u)
                            : __builtin_amdgcn_perm(d[4 - j], d[3 - j], 0x00010203u + 0x01010101u * (u32)(4 - c)));
      *(v4i *)(T + c * g.tpitch + w0) = o;
    }
  }
  wave_lds_fence();
}

// ---- products modulo x^N - 1 ONLY (Newton rounds, public key): ONE matrix instruction per tile distance -------------------------
// The wrapped terms a[i] s[k - i + N] (i > k) meet the SAME fragment as the unwrapped ones of distance d when their rows come from a
// copy of `a` moved up by P = 32 NT - N places (tools/peritem_mfma_model.py product_cyclic_registers): row kb of distance d is chunk
// kb - d of a (kb >= d) or chunk kb - d + NT of the moved copy (kb < d).  So the rows walk UP one lane per distance, and chunk NT - d of
// the moved copy ENTERS at row 0 -- two lanes -- from a byte image of that copy in the wave's LDS (byte P + i = a[i]; every lane of a
// half-wave reads the same 16 bytes: a broadcast).  One v_cndmask_b32_dpp per dword does both (vcc = lanes 0 and 32: the entering
// halves; every other lane its lower neighbour's dword -- which also cuts the seam at lane 32).  The distance-0 tile is taken whole
// (above its diagonal lie the wrapped terms inside the tile), so the P coefficients that chunk kb + 1 of the moved copy shares with
// chunk kb of `a` are cut from the rows of the LAST distance, where that chunk sits at row kb.  NT matrix instructions and one
// accumulator per plane where the split form takes 2 NT and two.
constexpr int PI_IMG = 1152;        // bytes between the images of two planes (P + 32 NT <= 1055)

static __device__ __forceinline__ v4i rows_up_enter(const v4i a, const v4i e) {
  int o0, o1, o2, o3;                                      // (untied, early-clobber outputs: the rows ping-pong between two register tuples, no moves)
  asm volatile(
      "s_mov_b32 vcc_lo, 1\n\ts_mov_b32 vcc_hi, 1\n\ts_nop 1\n\t"
      "v_cndmask_b32_dpp %0, %4, %8, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %1, %5, %9, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %2, %6, %10, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_cndmask_b32_dpp %3, %7, %11, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3])
      : "vcc");
  return (v4i){o0, o1, o2, o3};
}

// This lane's 16 bytes of a plane (chunk ch: coefficients 16 ch .. 16 ch + 15, zero at and beyond N) into the plane's image.
static __device__ __forceinline__ void pi_store_image(unsigned char *img, const PGeom &g, int ch, v4i bytes) {
  if (16 * ch < 32 * g.NT) *(v4i *)(img + (32 * g.NT - g.N) + 16 * ch) = bytes;     // (any alignment)
}

// NPL planes F[p] (this lane's unshifted chunk rows) whose images lie at img + p PI_IMG, against the fragments of T: C[p] = plane p's
// product modulo x^N - 1.  Fragment and entering rows are requested one trip ahead, unrolled by two so that the sets rotate without moves.
// The byte mask of the last distance (pi_product_cyc): rows <= NT - 2 lose their bytes 16 hh + j < P = 32 NT - N.  Once per item.
static __device__ __forceinline__ void pi_cyc_keep(const PGeom &g, int lane_, u32 (&keep)[4]) {
  const int lane = opaque(lane_), P = 32 * g.NT - g.N, u = (lane & 31) <= g.NT - 2 ? P - 16 * (lane >> 5) : 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int n = min(max(u - 4 * c, 0), 4);
    keep[c] = n >= 4 ? 0u : ~((1u << (8 * n)) - 1u);
  }
}

template <int NPL>
static __device__ __forceinline__ void pi_product_cyc(const v4i (&F)[NPL], const unsigned char *img, const u32 *T, const PGeom &g, int lane_,
                                                      const u32 (&keep)[4], v16i (&C)[NPL]) {
  const int lane = opaque(lane_), NT = g.NT;
  const int y0 = 32 * NT - 1 - (lane & 31) + 16 * (lane >> 5);
  const u32 *tb = T + (y0 & 3) * g.tpitch + (y0 >> 2);     // this lane's fragment of distance 0; distance d lies 8 d dwords below
  const unsigned char *eb = img + 32 * NT + 16 * (lane >> 5);                // chunk NT - d of the image: 32 d bytes below
  auto frag = [&](int d) {
    const u32 *p = tb - 8 * d;
    return (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
  };
  struct Rows { v4i e[NPL]; };
  auto enter = [&](int d) {
    Rows r;
#pragma unroll
    for (int p = 0; p < NPL; p++) r.e[p] = *(const v4i *)(eb - 32 * d + p * PI_IMG);
    return r;
  };
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  v4i A[NPL];
  const v4i w0 = frag(0);
  v4i w_a = frag(1), w_b;
  Rows e_a = enter(1), e_b;
#pragma unroll
  for (int p = 0; p < NPL; p++) {
    C[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[p], w0, zero, 0, 0, 0);
    A[p] = F[p];
  }
  auto trip = [&](const v4i &w, const Rows &e) {
#pragma unroll
    for (int p = 0; p < NPL; p++) {
      A[p] = rows_up_enter(A[p], e.e[p]);
      C[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], w, C[p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto last = [&](const v4i &w, const Rows &e) {           // distance NT - 1: rows <= NT - 2 lose their bytes 16 hh + j < P
#pragma unroll
    for (int p = 0; p < NPL; p++) {
      A[p] = and4(rows_up_enter(A[p], e.e[p]), keep);
      C[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[p], w, C[p], 0, 0, 0);
    }
  };
  int j = 1;
  for (; j + 1 <= NT - 2; j += 2) {
    w_b = frag(j + 1); e_b = enter(j + 1);
    trip(w_a, e_a);
    w_a = frag(j + 2); e_a = enter(j + 2);
    trip(w_b, e_b);
  }
  if (j <= NT - 2) {
    w_b = frag(j + 1); e_b = enter(j + 1);
    trip(w_a, e_a);
    last(w_b, e_b);
  } else {
    last(w_a, e_a);
  }
}

// Per wave: three natural-order periods of the ternary operand (the source of its reversed array; later the remainder of product 3
// for the comparison with h), then the reversed array.  No chunk matrix: the rows live in registers.
static __host__ __device__ inline size_t pi_reg_wave_bytes(const PGeom &g) { return pi_nat_bytes(g) + (size_t)16 * g.tpitch; }

// Reversed cyclic array (4 byte-shifted copies) of a ternary / int8 operand of which this lane holds chunk ch (the 16 bytes sv:
// coefficients 16 ch .. 16 ch + 15, zero at and beyond N; any assignment of chunks to lanes): three periods in natural order (period
// k starts at byte k N, any alignment: unaligned LDS stores), then T[c][w] = bytes rev[4w + c + j], rev[y] = s[(Y0 - y) mod N], as
// byte-swapped reads.  The last chunk of a period is stored whole: its zero tail lands on the next period's first bytes, which the
// NEXT store instruction writes (the LDS executes one wave's instructions in order).
static __device__ __forceinline__ void pi_build_array_ch(unsigned char *nat, u32 *T, const PGeom &g, int lane_, int ch, v4i sv) {
  const int N = g.N, Y0 = 32 * g.NT - 1, lane = opaque(lane_);
  const bool holds = 16 * ch < N;
#pragma unroll
  for (int k = 0; k < 3; k++)
    if (holds) *(v4i *)(nat + k * N + 16 * ch) = sv;
  if (ch < 4) *(v4i *)(nat + 3 * N + 16 * ch) = sv;        // N >= 64
  wave_lds_fence();
  // Word w of copy c holds bytes nat[A .. A+3] reversed, A = E - c, E = Y0 + 2N - 3 - 4w.  E & 3 is the same for every
  // lane, so the four copies of a word come from three ALIGNED dwords around E >> 2 with one byte permute each
  // (an unaligned LDS dword read costs several aligned ones).
  const u32 *D = (const u32 *)nat;
  const int e = __builtin_amdgcn_readfirstlane((Y0 + 2 * N - 3) & 3);
  u32 sel[4]; int dk[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int al = c <= e ? e - c : e - c + 4;                            // byte offset of A inside its dword
    dk[c] = c <= e ? 0 : -1;                                              // ... which is dword K or K - 1
    sel[c] = 0x00010203u + 0x01010101u * (u32)al;                         // bytes al+3, al+2, al+1, al of the pair (reversed)
  }
  // A lane makes FOUR consecutive words of all four copies per trip (one 16-byte store per copy) from six consecutive source dwords:
  // two trips cover a copy at N = 821 where word-per-lane trips took seven dependent LDS round trips.
  const int K0 = (Y0 + 2 * N - 3) >> 2;
  for (int w0 = 4 * lane; w0 < g.tpitch; w0 += 256) {      // tpitch is a multiple of 4 (and a copy's size of 16 bytes)
    int base = K0 - w0 - 4;                                // words w0 + j need dwords K0 - w0 - j - 1 .. K0 - w0 - j + 1
    base = base < 0 ? 0 : base;                            // (only pad words of a copy, which are never read, lie that far out)
    u32 d[6];
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = D[base + i];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      v4i o;
#pragma unroll
      for (int j = 0; j < 4; j++)
        o[j] = (int)(dk[c] == 0 ? __builtin_amdgcn_perm(d[5 - j], d[4 - j], sel[c]) : __builtin_amdgcn_perm(d[4 - j], d[3 - j], sel[c]));
      *(v4i *)(T + c * g.tpitch + w0) = o;
    }
  }
  wave_lds_fence();
}

// verifyKeysInputs (index.js:141-197) for one key pair per wave, ALL THREE products in one pass over the tile distances.
// Lane (r, hh) = 32 hh + r holds chunk ch = 2 r + hh (16 coefficients) of every operand row -- the layout the matrix instruction
// wants its A operand in -- so the rows go from HBM to the matrix cores through registers only.  Product 3 is taken as
// p (fq * g): ((p fq) mod q) * g = p (fq * g) modulo q for the low and the high half alike, so its rows are the SAME two digit
// planes of fq that product 1 multiplies, and the factor p goes into its epilogue.  The three planes (fq lo, fq hi, fp mod 3)
// are shifted once per distance and direction and meet the fragments of BOTH reversed arrays (f: products 1 and 2; g: product 3):
// ten matrix instructions per trip for 24 lane-shift instructions (two loops took 40), one set of planes, one loop prologue, fq
// read once.  Five accumulator pairs = 160 registers: two waves per SIMD.  Same device: 2.12-2.13 ms per 2^18 against 2.18-2.20 ms for
// the two-loop form at three waves per SIMD (products 1 + 2, then product 3 with its own planes), 2.09 against 2.18 J per launch.
static __host__ __device__ inline size_t pi_verify_wave_bytes(const PGeom &g) { return pi_nat_bytes(g) + (size_t)32 * g.tpitch; }

__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_verify_keys_m(
    PGeom g, u32 q, const int8_t *__restrict__ f, const int8_t *__restrict__ gg, const u16 *__restrict__ fq,
    const uint8_t *__restrict__ fp, const u16 *__restrict__ h, long B, u16 *__restrict__ quot_fq,
    u16 *__restrict__ rem_fq, uint8_t *__restrict__ quot_fp, uint8_t *__restrict__ rem_fp, u16 *__restrict__ quot_h,
    u16 *__restrict__ rem_h, uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *nat = lds + (size_t)wave * pi_verify_wave_bytes(g);
  u32 *Tf = (u32 *)(nat + pi_nat_bytes(g)), *Tg = Tf + 4 * g.tpitch;
  const int N = g.N, NT = g.NT;
  auto chunk_of = [](int ln) { return 2 * (ln & 31) + (ln >> 5); };      // this lane's chunk of every operand row
  auto index_of = [](int ln) { return 128 * (ln >> 5) + (ln & 31); };    // accumulator register i holds index 32 ((i&3) + 8 (i>>2)) + this
  const long item_step = (long)gridDim.x * PI_WAVES;
  // An item's operand rows are requested at the end of the PREVIOUS item's last epilogue (a wave that fetched them where it needs
  // them sat idle for a round trip to HBM per item) and stay in these registers across the loop back-edge.
  RawChunks<2> r_fq;
  RawChunks<1> r_f, r_fp, r_g;
  auto request_rows = [&](long it) {
    // (descriptors of ONE row: the lanes whose chunk lies beyond it -- chunks NT .. 63 -- read zeros instead of fetching the next items' rows)
    const long row = it * N;
    const int ch = chunk_of(opaque(lane));
    const AlignedSrc s_fq = aligned_src(fq + row, 2L * N), s_f = aligned_src(f + row, (long)N), s_fp = aligned_src(fp + row, (long)N),
                     s_g = aligned_src(gg + row, (long)N);
    r_fq = load_raw<2>(s_fq, s_fq.a0 + 32 * ch, 0); r_f = load_raw<1>(s_f, s_f.a0 + 16 * ch, 0);
    r_g = load_raw<1>(s_g, s_g.a0 + 16 * ch, 0); r_fp = load_raw<1>(s_fp, s_fp.a0 + 16 * ch, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request_rows((long)blockIdx.x * PI_WAVES + wave);
  [[maybe_unused]] int stamp_iter = -1;                    // -DNTRU_STAMPS: phase stamps of the first items (tools/phase_stamps_peritem.py)
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N;
    u32 fl = 0;
    stamp_iter++;
    STAMP(0);
    auto bytes_of = [&](const RawChunks<1> &rw, const void *p) {
      v4i v[1];
      shift_raw<1>(rw, __builtin_amdgcn_readfirstlane((int)((unsigned long long)p & 15)), v);
      return v[0];
    };
    // ternary operands: any negative byte is -1 (ValTernary), bytes at and beyond N are zero -- four bytes at a time
    auto ternary = [&](v4i v, const v4i &cmask) {
      v4i o;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const u32 w = (u32)(v[c] & cmask[c]);
        const u32 neg = (w >> 7) & 0x01010101u;            // 1 in every negative byte ...
        o[c] = (int)(w | neg * 0xFFu);                     // ... spread to 0xFF (no carries: the factors are below 2^8 and 2^25)
      }
      return o;
    };
    // ---- operands: the reversed arrays of f and g, the planes fq lo / fq hi / fp mod 3 (index.js:155-166)
    v4i F[3];
    {
      const int ch = chunk_of(opaque(lane));
      const v4i cmask = col_mask16(16 * ch, N);            // bytes of this lane's chunk that are below N
      pi_build_array_ch(nat, Tf, g, lane, ch, ternary(bytes_of(r_f, f + row), cmask));
      pi_build_array_ch(nat, Tg, g, lane, ch, ternary(bytes_of(r_g, gg + row), cmask));
      STAMP(1);                                            // the two reversed arrays
      u32 xq[8];
      {
        v4i v[2];
        shift_raw<2>(r_fq, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(fq + row) & 15)), v);
#pragma unroll
        for (int c = 0; c < 4; c++) { xq[c] = (u32)v[0][c]; xq[4 + c] = (u32)v[1][c]; }
      }
      pi_digits(xq, q, 1u, 16 * ch, N, F[0], F[1]);
      // fp mod 3 as the third plane.  A key's fp is already reduced: one wave-wide test (is any byte >= 3?) skips the byte-wise division
      union { v4i v; unsigned char c[16]; } u; u.v = bytes_of(r_fp, fp + row) & cmask;
      u32 big = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) big |= ((((u32)u.v[c] & 0x7F7F7F7Fu) + 0x7D7D7D7Du) | (u32)u.v[c]) & 0x80808080u;
      if (__ballot(big != 0) != 0) {
#pragma unroll
        for (int j = 0; j < 16; j++) u.c[j] = (unsigned char)((u32)u.c[j] % 3u);
      }
      F[2] = u.v;
    }
    // h is requested before the loop whose remainder it is compared with, as a natural-order row chunk (16 coefficients per lane);
    // the remainder gets into the same layout through the wave's LDS (the natural-order area is free again by then)
    const AlignedSrc s_h = aligned_src(h + row, 2L * N);
    const RawChunks<2> r_h = load_raw<2>(s_h, s_h.a0 + 32 * opaque(lane), 0);
    STAMP(2);                                              // the three planes in registers
    // ---- the loop: L1 / H1 (two planes) = fq * f, L2 / H2 = fp * f, L3 / H3 (two planes) = fq * g
    v16i L1[2], H1[2], L2, H2, L3[2], H3[2];
    {
      const int ln = opaque(lane);
      const int y0 = 32 * NT - 1 - (ln & 31) + 16 * (ln >> 5);
      const u32 *tb = Tf + (y0 & 3) * g.tpitch + (y0 >> 2);     // this lane's fragment of f at distance 0; g's lies 4 tpitch dwords above
      const int tstep = 4 * g.tpitch;
      int seam_up = ln == 32 ? 0 : -1, seam_dn = ln == 31 ? 0 : -1;
      asm volatile("" : "+v"(seam_up), "+v"(seam_dn));
      struct Fr { v4i f, g; };
      auto frag = [&](int d) {                             // |d| <= NT - 1; requests past the last step read the last fragment again
        d = d > NT - 1 ? NT - 1 : (d < 1 - NT ? 1 - NT : d);
        const u32 *p = tb - 8 * d, *p1 = p + tstep;
        Fr fr;
        fr.f = (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
        fr.g = (v4i){(int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
        return fr;
      };
      const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const Fr f0 = frag(0);
      Fr la = frag(1), ha = frag(-1), lb, hb;
      {                                                    // d = 0: split by the diagonal mask; the first term of every accumulator
        u32 mlow[4], mhigh[4];
        pi_diag_low_mask(ln, mlow);
#pragma unroll
        for (int c = 0; c < 4; c++) mhigh[c] = ~mlow[c];
        const v4i fl_ = and4(f0.f, mlow), fh_ = and4(f0.f, mhigh), gl_ = and4(f0.g, mlow), gh_ = and4(f0.g, mhigh);
#pragma unroll
        for (int p = 0; p < 2; p++) {
          L1[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[p], fl_, zero, 0, 0, 0);
          H1[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[p], fh_, zero, 0, 0, 0);
          L3[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[p], gl_, zero, 0, 0, 0);
          H3[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[p], gh_, zero, 0, 0, 0);
        }
        L2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[2], fl_, zero, 0, 0, 0);
        H2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[2], fh_, zero, 0, 0, 0);
      }
      v4i AL[3] = {F[0], F[1], F[2]}, AH[3] = {F[0], F[1], F[2]};
      auto trip = [&](const Fr &wl, const Fr &wh) {        // every shifted plane feeds two matrix instructions (the fp plane one)
#pragma unroll
        for (int p = 0; p < 2; p++) {
          AL[p] = rows_up(AL[p], seam_up);
          L1[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AL[p], wl.f, L1[p], 0, 0, 0);
          L3[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AL[p], wl.g, L3[p], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);               // the next plane's shifts issue under these matrix instructions
        }
        AL[2] = rows_up(AL[2], seam_up);
        L2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(AL[2], wl.f, L2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 2; p++) {
          AH[p] = rows_down(AH[p], seam_dn);
          H1[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AH[p], wh.f, H1[p], 0, 0, 0);
          H3[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AH[p], wh.g, H3[p], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        AH[2] = rows_down(AH[2], seam_dn);
        H2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(AH[2], wh.f, H2, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      int j = 1;
      for (; j + 1 < NT; j += 2) {
        lb = frag(j + 1); hb = frag(-(j + 1));
        trip(la, ha);
        la = frag(j + 2); ha = frag(-(j + 2));
        trip(lb, hb);
      }
      if (j < NT) trip(la, ha);
    }
    STAMP(3);                                              // the matrix loop
    const int kl = index_of(opaque(lane));
    {
      // stores through one-row descriptors: index k = 32 kb + r is a per-lane offset (128 hh + r) plus a compile-time
      // one per register, indices >= N fall outside the descriptor and are dropped -- no address arithmetic per store
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_fq + row, 2L * N), rs_q = rows_rsrc(quot_fq + row, 2L * N);
      // index.js:159: invalid iff the remainder has a non-zero coefficient above the constant one AND its constant one is not 1.
      // Register i of this lane holds index ko_i + kl: it exists iff ko_i < N - kl (one compare against a per-lane limit, ko_i a
      // constant); coefficient 0 is register 0 of lane 0.
      u32 any_hi = 0, c0 = 0;
      const int lim = N - kl;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const int lo = L1[0][i] + 128 * L1[1][i], hi = H1[0][i] + 128 * H1[1][i];
        const u32 rv = (u32)(lo + hi) & (q - 1);
        __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
        if (i == 0) { c0 = rv; any_hi |= kl == 0 ? 0u : rv; }
        else any_hi |= ko < lim ? rv : 0u;
      }
      const bool nz_hi = any_hi != 0, first_not_one = kl == 0 && c0 != 1;
      if (__ballot(nz_hi) != 0 && __ballot(first_not_one) != 0) fl |= NTRU_FLAG_INVALID_FQ;   // length !== 1 && [0] !== 1
    }
    STAMP(4);                                              // product 1's epilogue
    {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_fp + row, (long)N), rs_q = rows_rsrc(quot_fp + row, (long)N);
      u32 any_hi = 0, c0 = 0;                              // as for product 1
      const int lim = N - kl;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        // |L + H|, |H| <= 127 N (f is an int8, fp < 3): a multiple of 3 above that keeps the dividend non-negative
        const u32 x = (u32)(L2[i] + H2[i] + 3 * 131072), y = (u32)(3 * 131072 - H2[i]);
        const u32 rv = x % 3u, qv = y % 3u;
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)rv, rs_r, kl, ko, 0);
        __builtin_amdgcn_raw_buffer_store_b8((uint8_t)qv, rs_q, kl, ko, 0);
        if (i == 0) { c0 = rv; any_hi |= kl == 0 ? 0u : rv; }
        else any_hi |= ko < lim ? rv : 0u;
      }
      const bool nz_hi = any_hi != 0, first_not_one = kl == 0 && c0 != 1;
      if (__ballot(nz_hi) != 0 && __ballot(first_not_one) != 0) fl |= NTRU_FLAG_INVALID_FP;
    }
    STAMP(5);                                              // product 2's epilogue
    {
      // product 3 = p (fq * g) mod q (index.js:155,164): the factor p = 3 is applied here, to the low and the high half alike
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_h + row, 2L * N), rs_q = rows_rsrc(quot_h + row, 2L * N);
      u16 *remx = (u16 *)nat;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const int lo = L3[0][i] + 128 * L3[1][i], hi = H3[0][i] + 128 * H3[1][i];
        const u32 rv = (u32)(3 * (lo + hi)) & (q - 1);
        __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - 3 * hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
        remx[ko + kl] = (u16)rv;                           // ko + kl <= 1151: inside the area for every N (pi_nat_bytes)
      }
      if (item + item_step < B) request_rows(item + item_step);   // the next item's rows: in flight from here on (the accumulators are dead)
      wave_lds_fence();
      STAMP(6);                                            // product 3's result stores issued
      // index.js:165: h[k] must equal the remainder for every k below h's trimmed length
      v4i hc[2];
      shift_raw<2>(r_h, __builtin_amdgcn_readfirstlane(s_h.a0), hc);
      const int i0 = 16 * opaque(lane);
      // Per lane two 16-bit sets in a SPLIT layout (coefficient i0 + 2 c in bit c, i0 + 2 c + 1 in bit 16 + c: what one packed
      // 16-bit minimum and one shift-or per dword give): nz = h is non-zero there, df = h differs from the remainder there;
      // coefficients at and beyond N are cut off the sets, not off the data.  h is invalid iff its FIRST difference from the
      // remainder lies at or below its LAST non-zero coefficient (index.js:165 compares below h's trimmed length); both are
      // found on the scalar side from one ballot and one v_readlane each.
      u32 nz = 0, df = 0;
      if (i0 < 32 * NT) {
        const v4i rc0 = *(const v4i *)(nat + 2 * i0), rc1 = *(const v4i *)(nat + 2 * i0 + 16);
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const u32 hx = (u32)(c < 4 ? hc[0][c] : hc[1][c - 4]), rx = (u32)(c < 4 ? rc0[c] : rc1[c - 4]);
          nz |= as_u32(__builtin_elementwise_min(as_pair(hx), (u16x2){1, 1})) << c;
          df |= as_u32(__builtin_elementwise_min(as_pair(hx ^ rx), (u16x2){1, 1})) << c;
        }
        const int left = N - i0;                           // coefficients of this lane's chunk that exist
        const u32 ne = left >= 16 ? 0xFFu : (1u << ((left + 1) >> 1)) - 1u, no = left >= 16 ? 0xFFu : (1u << (left >> 1)) - 1u;
        const u32 valid = left <= 0 ? 0u : (ne | no << 16);
        nz &= valid; df &= valid;
      }
      auto last_of = [](u32 m) {                           // highest coefficient (0..15) of a non-empty split set
        const u32 ev = m & 0xFFFFu, od = m >> 16;
        const int te = ev ? 2 * (31 - __builtin_clz(ev)) : -1, to = od ? 2 * (31 - __builtin_clz(od)) + 1 : -1;
        return te > to ? te : to;
      };
      auto first_of = [](u32 m) {                          // lowest coefficient of a non-empty split set
        const u32 ev = m & 0xFFFFu, od = m >> 16;
        const int fe = ev ? 2 * __builtin_ctz(ev) : 64, fo = od ? 2 * __builtin_ctz(od) + 1 : 64;
        return fe < fo ? fe : fo;
      };
      const unsigned long long has = __ballot(nz != 0), dif = __ballot(df != 0);
      if (dif) {
        const int lf = __builtin_ctzll(dif);
        const int first_diff = 16 * lf + first_of((u32)__builtin_amdgcn_readlane((int)df, lf));
        int top = 0;                                       // the zero polynomial has trimmed length 1: index 0 is compared
        if (has) {
          const int lt = 63 - __builtin_clzll(has);
          top = 16 * lt + last_of((u32)__builtin_amdgcn_readlane((int)nz, lt));
        }
        if (first_diff <= top) fl |= NTRU_FLAG_INVALID_H;
      }
    }
    if (lane == 0) flags[item] = (uint8_t)fl;
    wave_lds_fence();
    STAMP(7);                                              // the comparison with h
  }
}

#ifdef NTRU_EXPERIMENTS
// ---- EXPERIMENT (kernel path 12, `make experiments` only): the same kernel on the 16-row tile, v_mfma_i32_16x16x64_i8 (K = 64: TWO tile
// distances per instruction, two 16-column halves).  Bit-exact, measured SLOWER: 2.49 against 2.07 ms per 2^18 key pairs on one device
// (EXPERIMENTS.md round 5, item 9): its loop issues 0.73x the matrix clocks but 1.5x the vector instructions around them (17-20 k against
// 12.5 k clocks per key pair), and its result stores leave as 32-byte segments, four per instruction (epilogues 11 k against 6.4 k). ----
// With 32 rows per instruction only NT - |d| of them carry data (41 % at N = 821) and the instruction costs what it costs whatever its
// rows hold (bench_micro/mfma_rows_energy: 18.5 nJ with 32 live rows, 16.8 with 13, 15.3 with none).  With 16 rows the output tiles
// form two row groups (kb = 16 g + m) and a (group, distance pair) is only issued while it holds rows: per plane 78 instructions of 16
// clocks instead of 52 of 32 at NT = 26, HALF the matrix clocks for NT <= 16 (tools/peritem_mfma_model.py replays the scheme lane for
// lane).  Lane l = (m, kq), m = l & 15, kq = l >> 4: kq 0, 1 hold the two 16-byte K halves of the pair's first distance, kq 2, 3 of its
// second, i.e. of the chunk one row further.
//   low walk, pairs (0, 1), (2, 3), ...:   A = chunk kb - d - (kq >> 1); a pair step moves every 16-lane row by two rows (row_shr:2),
//                                           group 1 takes its two entering rows from group 0's top two (row_ror:2 as the DPP `old`);
//   high walk, pairs (0h, -1), (-2, -3), ..: A = chunk kb + e + (kq >> 1); row_shl:2, group 0 takes its entering rows from group 1.
// No seam masks (row operations stay inside 16 lanes).  The planes reach this layout through a natural-order byte image in the LDS
// (chunks -1 .. 33, zero outside the operand): six 16-byte reads per walk.  Accumulators [low / high][group][column half] x 4 registers:
// 160 for the five plane products, as before.  Result index of register j of lane (n, rq): 32 (16 g + 4 rq + j) + 16 c + n.
constexpr int PI16_PLANE = 35 * 32;                         // bytes of one plane's image: chunks -1 .. 33
static __host__ __device__ inline size_t pi_verify16_area(const PGeom &g) {
  const size_t a = pi_nat_bytes(g), b = (3 * PI16_PLANE + 15) & ~(size_t)15;
  return a > b ? a : b;
}
static __host__ __device__ inline size_t pi_verify16_wave_bytes(const PGeom &g) { return pi_verify16_area(g) + (size_t)32 * g.tpitch; }

__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_verify_keys_m16(
    PGeom g, u32 q, const int8_t *__restrict__ f, const int8_t *__restrict__ gg, const u16 *__restrict__ fq,
    const uint8_t *__restrict__ fp, const u16 *__restrict__ h, long B, u16 *__restrict__ quot_fq,
    u16 *__restrict__ rem_fq, uint8_t *__restrict__ quot_fp, uint8_t *__restrict__ rem_fp, u16 *__restrict__ quot_h,
    u16 *__restrict__ rem_h, uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *nat = lds + (size_t)wave * pi_verify16_wave_bytes(g);
  u32 *Tf = (u32 *)(nat + pi_verify16_area(g)), *Tg = Tf + 4 * g.tpitch;
  const int N = g.N, NT = g.NT;
  const long item_step = (long)gridDim.x * PI_WAVES;
  RawChunks<2> r_fq;
  RawChunks<1> r_f, r_fp, r_g;
  auto request_rows = [&](long it) {                       // natural lane order: lane l holds coefficients 16 l .. 16 l + 15 (one-row descriptors)
    const long row = it * N;
    const int ln = opaque(lane);
    const AlignedSrc s_fq = aligned_src(fq + row, 2L * N), s_f = aligned_src(f + row, (long)N), s_fp = aligned_src(fp + row, (long)N),
                     s_g = aligned_src(gg + row, (long)N);
    r_fq = load_raw<2>(s_fq, s_fq.a0 + 32 * ln, 0); r_f = load_raw<1>(s_f, s_f.a0 + 16 * ln, 0);
    r_g = load_raw<1>(s_g, s_g.a0 + 16 * ln, 0); r_fp = load_raw<1>(s_fp, s_fp.a0 + 16 * ln, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request_rows((long)blockIdx.x * PI_WAVES + wave);
  [[maybe_unused]] int stamp_iter = -1;
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N;
    u32 fl = 0;
    stamp_iter++;
    STAMP(0);
    auto bytes_of = [&](const RawChunks<1> &rw, const void *p) {
      v4i v[1];
      shift_raw<1>(rw, __builtin_amdgcn_readfirstlane((int)((unsigned long long)p & 15)), v);
      return v[0];
    };
    auto ternary = [&](v4i v, const v4i &cmask) {          // any negative byte is -1 (ValTernary), bytes at and beyond N are zero
      v4i o;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const u32 w = (u32)(v[c] & cmask[c]);
        o[c] = (int)(w | ((w >> 7) & 0x01010101u) * 0xFFu);
      }
      return o;
    };
    // ---- operands: the reversed arrays of f and g; then the planes fq lo / fq hi / fp mod 3 as natural-order byte images
    {
      const int ln = opaque(lane);
      const v4i cmask = col_mask16(16 * ln, N);
      pi_build_array_ch(nat, Tf, g, lane, ln, ternary(bytes_of(r_f, f + row), cmask));
      pi_build_array_ch(nat, Tg, g, lane, ln, ternary(bytes_of(r_g, gg + row), cmask));
      STAMP(1);
      u32 xq[8];
      {
        v4i v[2];
        shift_raw<2>(r_fq, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(fq + row) & 15)), v);
#pragma unroll
        for (int c = 0; c < 4; c++) { xq[c] = (u32)v[0][c]; xq[4 + c] = (u32)v[1][c]; }
      }
      v4i d0, d1;
      pi_digits(xq, q, 1u, 16 * ln, N, d0, d1);
      union { v4i v; unsigned char c[16]; } u; u.v = bytes_of(r_fp, fp + row) & cmask;
      u32 big = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) big |= ((((u32)u.v[c] & 0x7F7F7F7Fu) + 0x7D7D7D7Du) | (u32)u.v[c]) & 0x80808080u;
      if (__ballot(big != 0) != 0) {
#pragma unroll
        for (int j = 0; j < 16; j++) u.c[j] = (unsigned char)((u32)u.c[j] % 3u);
      }
      // chunk ch lies at byte 32 (ch + 1) of its plane: the 64 lanes fill chunks 0 .. 31 (zeros at and beyond N); chunks -1, 32, 33
      // were overwritten by the arrays' staging and are zeroed again (18 lanes)
      *(v4i *)(nat + 0 * PI16_PLANE + 32 + 16 * ln) = d0;
      *(v4i *)(nat + 1 * PI16_PLANE + 32 + 16 * ln) = d1;
      *(v4i *)(nat + 2 * PI16_PLANE + 32 + 16 * ln) = u.v;
      if (ln < 18) {
        const int pl = ln / 6, k6 = ln - 6 * pl;           // per plane: bytes 0 .. 31 and 1056 .. 1119 = six 16-byte pieces
        *(v4i *)(nat + pl * PI16_PLANE + (k6 < 2 ? 16 * k6 : 1056 + 16 * (k6 - 2))) = (v4i){0, 0, 0, 0};
      }
      wave_lds_fence();
    }
    STAMP(2);
    // ---- the two walks.  acc1 / acc3: [plane][low, high][group][column half] for fq * f / fq * g; acc2: fp * f
    v4i acc1[2][2][2][2], acc2[2][2][2], acc3[2][2][2][2];
    {
      const int ln = opaque(lane), n = ln & 15, kq = ln >> 4;
      const v4i zero4 = {0, 0, 0, 0};
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b_ = 0; b_ < 2; b_++)
#pragma unroll
          for (int c = 0; c < 2; c++) {
            acc2[a][b_][c] = zero4;
#pragma unroll
            for (int p = 0; p < 2; p++) { acc1[p][a][b_][c] = zero4; acc3[p][a][b_][c] = zero4; }
          }
      // this lane's fragment of f, column half 0, distance 0 (half 1: 4 dwords below; distance d: 8 d dwords below; g: 4 tpitch dwords above)
      const int y0 = 32 * NT - 1 - n + 16 * (kq & 1);
      const u32 *tb = Tf + (y0 & 3) * g.tpitch + (y0 >> 2);
      const int tstep = 4 * g.tpitch;
      struct Fr { v4i f[2], g[2]; };                       // fragments of one distance pair: [column half]
      const int pairs = (NT + 1) >> 1;
      auto walk = [&](auto high_tag) {
        constexpr bool HIGH = decltype(high_tag)::value;
        const u32 *tw = HIGH ? tb + 8 * (kq >> 1) : tb - 8 * (kq >> 1);          // the lane's second-distance offset folded in
        auto frag = [&](int pr) {
          pr = pr < pairs ? pr : pairs - 1;                // requests past the last pair read it again (never used)
          const u32 *p = HIGH ? tw + 16 * pr : tw - 16 * pr;
          Fr fr;
#pragma unroll
          for (int c = 0; c < 2; c++) {
            const u32 *pc = p - 4 * c, *pg = pc + tstep;
            fr.f[c] = (v4i){(int)pc[0], (int)pc[1], (int)pc[2], (int)pc[3]};
            fr.g[c] = (v4i){(int)pg[0], (int)pg[1], (int)pg[2], (int)pg[3]};
          }
          return fr;
        };
        // rows of the planes in the walk's layout: chunk 16 g + m -+ (kq >> 1), K half kq & 1
        v4i A[3][2];
        {
          const int chunk = (ln & 15) + (HIGH ? (kq >> 1) : -(kq >> 1));
#pragma unroll
          for (int p = 0; p < 3; p++)
#pragma unroll
            for (int gr = 0; gr < 2; gr++) A[p][gr] = *(const v4i *)(nat + p * PI16_PLANE + 32 * (16 * gr + chunk + 1) + 16 * (kq & 1));
        }
        Fr fa = frag(0), fb;
        {                                                  // the pair with distance 0: only its low (high) part counts on lanes kq < 2:
#pragma unroll                                             // byte j of column half c is LOW iff 16 c + n >= 16 (kq & 1) + j
          for (int c = 0; c < 2; c++)
#pragma unroll
            for (int w = 0; w < 4; w++) {
              const int cnt = min(max(16 * c + n - 16 * (kq & 1) + 1 - 4 * w, 0), 4);
              const u32 ml = cnt >= 4 ? 0xFFFFFFFFu : ((1u << (8 * cnt)) - 1u);
              const u32 keep = kq >= 2 ? 0xFFFFFFFFu : (HIGH ? ~ml : ml);
              fa.f[c][w] = (int)((u32)fa.f[c][w] & keep); fa.g[c][w] = (int)((u32)fa.g[c][w] & keep);
            }
        }
        auto pair_step = [&](const Fr &w, auto g0_tag, auto g1_tag) {
          constexpr bool G0 = decltype(g0_tag)::value, G1 = decltype(g1_tag)::value;
#pragma unroll
          for (int p = 0; p < 3; p++) {
#pragma unroll
            for (int gr = 0; gr < 2; gr++) {
              if ((gr == 0 && !G0) || (gr == 1 && !G1)) continue;
#pragma unroll
              for (int c = 0; c < 2; c++) {
                if (p < 2) {
                  acc1[p][HIGH][gr][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p][gr], w.f[c], acc1[p][HIGH][gr][c], 0, 0, 0);
                  acc3[p][HIGH][gr][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p][gr], w.g[c], acc3[p][HIGH][gr][c], 0, 0, 0);
                } else {
                  acc2[HIGH][gr][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[p][gr], w.f[c], acc2[HIGH][gr][c], 0, 0, 0);
                }
              }
            }
            // the plane's rows move by two: towards higher rows in the low walk (group 1 takes group 0's top two), towards lower rows in
            // the high walk (group 0 takes group 1's bottom two); a group that holds no rows any more is not moved
#pragma unroll
            for (int w4 = 0; w4 < 4; w4++) {
              if (!HIGH) {
                if (G0 && G1) {
                  const int enter = __builtin_amdgcn_update_dpp(0, A[p][0][w4], 0x122, 0xf, 0xf, false);       // row_ror:2
                  A[p][1][w4] = __builtin_amdgcn_update_dpp(enter, A[p][1][w4], 0x112, 0xf, 0xf, false);       // row_shr:2, rows 0, 1 keep `enter`
                } else if (G1) {
                  A[p][1][w4] = __builtin_amdgcn_update_dpp(0, A[p][1][w4], 0x112, 0xf, 0xf, true);
                }
                if (G0) A[p][0][w4] = __builtin_amdgcn_update_dpp(0, A[p][0][w4], 0x112, 0xf, 0xf, true);      // zeros enter
              } else {
                if (G0 && G1) {
                  const int enter = __builtin_amdgcn_update_dpp(0, A[p][1][w4], 0x12E, 0xf, 0xf, false);       // row_ror:14: rows 14, 15 <- rows 0, 1
                  A[p][0][w4] = __builtin_amdgcn_update_dpp(enter, A[p][0][w4], 0x102, 0xf, 0xf, false);       // row_shl:2, rows 14, 15 keep `enter`
                } else if (G0) {
                  A[p][0][w4] = __builtin_amdgcn_update_dpp(0, A[p][0][w4], 0x102, 0xf, 0xf, true);
                }
                if (G1) A[p][1][w4] = __builtin_amdgcn_update_dpp(0, A[p][1][w4], 0x102, 0xf, 0xf, true);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        // pairs [0, n_both) with both groups, then [n_both, pairs) with the one that still holds rows
        //   low:  group 0 holds rows while 2 pr <= 15; group 1 (NT > 16) to the end
        //   high: group 0 to the end; group 1 holds rows while 16 + 2 pr <= NT - 1
        const int n_both = NT <= 16 ? 0 : (HIGH ? (NT - 15) >> 1 : 8);
        auto segment = [&](int first, int last, auto g0_tag, auto g1_tag) {       // pairs first .. last - 1; fa holds pair `first`
          int pr = first;
          for (; pr + 1 < last; pr += 2) {
            fb = frag(pr + 1);
            pair_step(fa, g0_tag, g1_tag);
            fa = frag(pr + 2);
            pair_step(fb, g0_tag, g1_tag);
          }
          if (pr < last) {
            fb = frag(pr + 1);
            pair_step(fa, g0_tag, g1_tag);
            fa = fb;
          }
        };
        using T_ = std::true_type; using F_ = std::false_type;
        if (NT <= 16) segment(0, pairs, T_{}, F_{});
        else {
          const int nb = n_both < pairs ? n_both : pairs;
          segment(0, nb, T_{}, T_{});
          if (HIGH) segment(nb, pairs, T_{}, F_{}); else segment(nb, pairs, F_{}, T_{});
        }
      };
      walk(std::false_type{});
      walk(std::true_type{});
    }
    STAMP(3);
    // h is requested here (the loop has no register to spare for it): the three epilogues are its flight time
    const AlignedSrc s_h = aligned_src(h + row, 2L * N);
    const RawChunks<2> r_h = load_raw<2>(s_h, s_h.a0 + 32 * opaque(lane), 0);
    const int kl = 128 * (opaque(lane) >> 4) + (opaque(lane) & 15);       // register j of (group g, column half c) holds index 512 g + 32 j + 16 c + kl
    {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_fq + row, 2L * N), rs_q = rows_rsrc(quot_fq + row, 2L * N);
      u32 any_hi = 0, c0 = 0;
      const int lim = N - kl;
#pragma unroll
      for (int gr = 0; gr < 2; gr++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int ko = 512 * gr + 32 * j + 16 * c;
            const int lo = acc1[0][0][gr][c][j] + 128 * acc1[1][0][gr][c][j], hi = acc1[0][1][gr][c][j] + 128 * acc1[1][1][gr][c][j];
            const u32 rv = (u32)(lo + hi) & (q - 1);
            __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
            __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
            if (ko == 0) { c0 = rv; any_hi |= kl == 0 ? 0u : rv; }
            else any_hi |= ko < lim ? rv : 0u;
          }
      const bool nz_hi = any_hi != 0, first_not_one = kl == 0 && c0 != 1;
      if (__ballot(nz_hi) != 0 && __ballot(first_not_one) != 0) fl |= NTRU_FLAG_INVALID_FQ;   // length !== 1 && [0] !== 1
    }
    STAMP(4);
    {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_fp + row, (long)N), rs_q = rows_rsrc(quot_fp + row, (long)N);
      u32 any_hi = 0, c0 = 0;
      const int lim = N - kl;
#pragma unroll
      for (int gr = 0; gr < 2; gr++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int ko = 512 * gr + 32 * j + 16 * c;
            const u32 x = (u32)(acc2[0][gr][c][j] + acc2[1][gr][c][j] + 3 * 131072), y = (u32)(3 * 131072 - acc2[1][gr][c][j]);
            const u32 rv = x % 3u, qv = y % 3u;
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)rv, rs_r, kl, ko, 0);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)qv, rs_q, kl, ko, 0);
            if (ko == 0) { c0 = rv; any_hi |= kl == 0 ? 0u : rv; }
            else any_hi |= ko < lim ? rv : 0u;
          }
      const bool nz_hi = any_hi != 0, first_not_one = kl == 0 && c0 != 1;
      if (__ballot(nz_hi) != 0 && __ballot(first_not_one) != 0) fl |= NTRU_FLAG_INVALID_FP;
    }
    STAMP(5);
    {
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem_h + row, 2L * N), rs_q = rows_rsrc(quot_h + row, 2L * N);
      u16 *remx = (u16 *)nat;
#pragma unroll
      for (int gr = 0; gr < 2; gr++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int ko = 512 * gr + 32 * j + 16 * c;
            const int lo = acc3[0][0][gr][c][j] + 128 * acc3[1][0][gr][c][j], hi = acc3[0][1][gr][c][j] + 128 * acc3[1][1][gr][c][j];
            const u32 rv = (u32)(3 * (lo + hi)) & (q - 1);
            __builtin_amdgcn_raw_buffer_store_b16((u16)rv, rs_r, 2 * kl, 2 * ko, 0);
            __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)(0 - 3 * hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
            remx[ko + kl] = (u16)rv;                       // ko + kl <= 624 + 399: inside the area
          }
      if (item + item_step < B) request_rows(item + item_step);
      wave_lds_fence();
      STAMP(6);
      // index.js:165: h[k] must equal the remainder for every k below h's trimmed length
      v4i hc[2];
      shift_raw<2>(r_h, __builtin_amdgcn_readfirstlane(s_h.a0), hc);
      const int i0 = 16 * opaque(lane);
      // Per lane two 16-bit sets in a SPLIT layout (coefficient i0 + 2 c in bit c, i0 + 2 c + 1 in bit 16 + c: what one packed
      // 16-bit minimum and one shift-or per dword give): nz = h is non-zero there, df = h differs from the remainder there;
      // coefficients at and beyond N are cut off the sets, not off the data.  h is invalid iff its FIRST difference from the
      // remainder lies at or below its LAST non-zero coefficient (index.js:165 compares below h's trimmed length); both are
      // found on the scalar side from one ballot and one v_readlane each.
      u32 nz = 0, df = 0;
      if (i0 < 32 * NT) {
        const v4i rc0 = *(const v4i *)(nat + 2 * i0), rc1 = *(const v4i *)(nat + 2 * i0 + 16);
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const u32 hx = (u32)(c < 4 ? hc[0][c] : hc[1][c - 4]), rx = (u32)(c < 4 ? rc0[c] : rc1[c - 4]);
          nz |= as_u32(__builtin_elementwise_min(as_pair(hx), (u16x2){1, 1})) << c;
          df |= as_u32(__builtin_elementwise_min(as_pair(hx ^ rx), (u16x2){1, 1})) << c;
        }
        const int left = N - i0;                           // coefficients of this lane's chunk that exist
        const u32 ne = left >= 16 ? 0xFFu : (1u << ((left + 1) >> 1)) - 1u, no = left >= 16 ? 0xFFu : (1u << (left >> 1)) - 1u;
        const u32 valid = left <= 0 ? 0u : (ne | no << 16);
        nz &= valid; df &= valid;
      }
      auto last_of = [](u32 m) {                           // highest coefficient (0..15) of a non-empty split set
        const u32 ev = m & 0xFFFFu, od = m >> 16;
        const int te = ev ? 2 * (31 - __builtin_clz(ev)) : -1, to = od ? 2 * (31 - __builtin_clz(od)) + 1 : -1;
        return te > to ? te : to;
      };
      auto first_of = [](u32 m) {                          // lowest coefficient of a non-empty split set
        const u32 ev = m & 0xFFFFu, od = m >> 16;
        const int fe = ev ? 2 * __builtin_ctz(ev) : 64, fo = od ? 2 * __builtin_ctz(od) + 1 : 64;
        return fe < fo ? fe : fo;
      };
      const unsigned long long has = __ballot(nz != 0), dif = __ballot(df != 0);
      if (dif) {
        const int lf = __builtin_ctzll(dif);
        const int first_diff = 16 * lf + first_of((u32)__builtin_amdgcn_readlane((int)df, lf));
        int top = 0;                                       // the zero polynomial has trimmed length 1: index 0 is compared
        if (has) {
          const int lt = 63 - __builtin_clzll(has);
          top = 16 * lt + last_of((u32)__builtin_amdgcn_readlane((int)nz, lt));
        }
        if (first_diff <= top) fl |= NTRU_FLAG_INVALID_H;
      }
    }
    if (lane == 0) flags[item] = (uint8_t)fl;
    wave_lds_fence();
    STAMP(7);
  }
}
#endif   // NTRU_EXPERIMENTS (verify_keys on the 16-row tile)

// One per-item product on the matrix cores: ((mul a) mod q) * s modulo x^N - 1 and q, a < 2^16 per item, s ternary per item:
// generatePublicKeyH (index.js:72-79, mul = p), whose quotient nobody asks for: pi_product_cyc, one matrix instruction per tile
// distance and digit plane.  ONE: a single int8 digit plane (q <= 256).
template <bool ONE>
__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(ONE ? 4 : 3, 4))) void k_product_tern_m(
    PGeom g, u32 q, u32 mul, const u16 *__restrict__ a, const int8_t *__restrict__ s, long B, u16 *__restrict__ rem) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int NPL = ONE ? 1 : 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *nat = lds + (size_t)wave * pi_reg_wave_bytes(g);
  u32 *T = (u32 *)(nat + pi_nat_bytes(g));
  const int N = g.N;
  auto chunk_of = [](int ln) { return 2 * (ln & 31) + (ln >> 5); };
  // The operands of the NEXT item are requested as soon as this item's are in registers (the round trip to HBM runs under the matrix
  // loops and the result stores instead of in front of every item).
  const long item_step = (long)gridDim.x * PI_WAVES;
  RawChunks<2> ra;
  RawChunks<1> rs;
  auto request = [&](long it) {
    const long rw = it * N;                                // (one-row descriptors: see k_verify_keys_m)
    const int ch = chunk_of(opaque(lane));
    const AlignedSrc sa = aligned_src(a + rw, 2L * N), ss = aligned_src(s + rw, (long)N);
    ra = load_raw<2>(sa, sa.a0 + 32 * ch, 0);
    rs = load_raw<1>(ss, ss.a0 + 16 * ch, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request((long)blockIdx.x * PI_WAVES + wave);
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N;
    v4i F[NPL];
    {
      const int ch = chunk_of(opaque(lane));
      v4i va[2], vs[1];
      shift_raw<2>(ra, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(a + row) & 15)), va);
      shift_raw<1>(rs, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(s + row) & 15)), vs);
      if (item + item_step < B) request(item + item_step);
      u32 xa[8];
#pragma unroll
      for (int c = 0; c < 4; c++) { xa[c] = (u32)va[0][c]; xa[4 + c] = (u32)va[1][c]; }
      pi_build_array_half(nat, T, g, lane, ch, pi_ternary(vs[0], col_mask16(16 * ch, N)));     // (its last fence: nat is free for the planes' images)
      v4i o0, o1;
      pi_digits(xa, q, mul, 16 * ch, N, o0, o1);
      F[0] = o0;
      pi_store_image(nat, g, ch, o0);
      if (!ONE) { F[NPL - 1] = o1; pi_store_image(nat + PI_IMG, g, ch, o1); }
      wave_lds_fence();
    }
    v16i C[NPL];
    u32 keep[4];
    pi_cyc_keep(g, lane, keep);
    pi_product_cyc<NPL>(F, nat, T, g, lane, keep, C);
    {
      const int ln = opaque(lane), kl = 128 * (ln >> 5) + (ln & 31);     // see k_verify_keys_m: indices >= N are dropped
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem + row, 2L * N);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const int c = C[0][i] + (ONE ? 0 : 128 * C[NPL - 1][i]);
        __builtin_amdgcn_raw_buffer_store_b16((u16)((u32)c & (q - 1)), rs_r, 2 * kl, 2 * ko, 0);
      }
    }
    wave_lds_fence();
  }
}

// One Newton round of the key inversion (polyInv, index.js:499-506) per item in ONE kernel, in its lifted form: v is right modulo
// 2^kb (kb <= 7 bits, the schedule of ntru_invert_key_batch_dev), f v = 1 + 2^kb e, and v <- v - 2^kb (e v mod 2^(m - kb)) is right
// modulo 2^m, m <= 2 kb.  Both products run on one int8 digit plane with the chunk rows in registers: f (x) v with f's reversed
// array and v (< 128) as the rows; then e -- at most kb bits per coefficient -- goes from the accumulator layout through a
// natural-order byte image in the LDS into the row layout (ONE read per lane), v's residues modulo 2^(m - kb) become the reversed
// array in f's place, and the second product's epilogue lifts v on the item's own row.
__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_newton_round_m(
    PGeom g, u32 kb, u32 m, const int8_t *__restrict__ f, u16 *v, long B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *nat = lds + (size_t)wave * pi_reg_wave_bytes(g);
  u32 *T = (u32 *)(nat + pi_nat_bytes(g));
  const int N = g.N;
  const u32 mr = 1u << m, me = 1u << (m - kb);
  auto chunk_of = [](int ln) { return 2 * (ln & 31) + (ln >> 5); };
  const long item_step = (long)gridDim.x * PI_WAVES;
  RawChunks<2> rv;
  RawChunks<1> rf;
  auto request = [&](long it) {
    const long rw = it * N;
    const int ch = chunk_of(opaque(lane));
    const AlignedSrc sv = aligned_src(v + rw, 2L * N), sf = aligned_src(f + rw, (long)N);
    rv = load_raw<2>(sv, sv.a0 + 32 * ch, 0);
    rf = load_raw<1>(sf, sf.a0 + 16 * ch, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request((long)blockIdx.x * PI_WAVES + wave);
  [[maybe_unused]] int stamp_iter = -1;                    // -DNTRU_STAMPS: phase stamps of the first items (tools/phase_stamps_peritem.py)
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N;
    stamp_iter++;
    STAMP(0);
    v4i b0;                                                // v modulo 2^(m - kb), centred: the second product's Toeplitz operand
    v4i F[1];
    {
      const int ch = chunk_of(opaque(lane));
      v4i va[2], vf[1];
      shift_raw<2>(rv, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(v + row) & 15)), va);
      shift_raw<1>(rf, __builtin_amdgcn_readfirstlane((int)((unsigned long long)(f + row) & 15)), vf);
      u32 xv[8];
#pragma unroll
      for (int c = 0; c < 4; c++) { xv[c] = (u32)va[0][c]; xv[4 + c] = (u32)va[1][c]; }
      v4i o1;
      pi_digits(xv, me, 1u, 16 * ch, N, b0, o1);
      pi_digits(xv, 256u, 1u, 16 * ch, N, F[0], o1);      // v < 2^kb <= 128: its own centred representative modulo 256
      STAMP(1);                                            // operands arrived, digits
      pi_build_array_half(nat, T, g, lane, ch, pi_ternary(vf[0], col_mask16(16 * ch, N)));       // (its last fence: nat is free for v's image)
      pi_store_image(nat, g, ch, F[0]);
      wave_lds_fence();
    }
    STAMP(2);                                              // f's reversed array, v's image
    v16i C[1];
    u32 keep[4];
    pi_cyc_keep(g, lane, keep);
    pi_product_cyc<1>(F, nat, T, g, lane, keep, C);        // f v
    STAMP(3);
    if (item + item_step < B) request(item + item_step);   // the next item's rows (nobody lifts them before this wave does): a product ahead
    {
      const int ln = opaque(lane), kl = 128 * (ln >> 5) + (ln & 31), ch = chunk_of(ln);
      u32 e[16];                                           // e = (f v - 1) / 2^kb modulo 2^(m - kb): bits kb .. m - 1 of f v - 1
#pragma unroll
      for (int i = 0; i < 16; i++)
        e[i] = __builtin_amdgcn_ubfe((u32)C[0][i] - (i == 0 && kl == 0 ? 1u : 0u), kb, m - kb);
      pi_build_array_half(nat, T, g, lane, ch, b0);          // (over f's array; its last fence orders the reads of nat before the writes below)
      unsigned char *img = nat + (32 * g.NT - N);          // e's image for the entering rows IS its natural-order byte image, P bytes up
#pragma unroll
      for (int i = 0; i < 16; i++) img[32 * ((i & 3) + 8 * (i >> 2)) + kl] = (unsigned char)e[i];     // index <= 1023 + 31
      wave_lds_fence();
      F[0] = *(const v4i *)(img + 16 * ch) & col_mask16(16 * ch, N);       // (any alignment; bytes at and beyond N: whatever the product left there)
    }
    // v in the accumulator layout, for the lift (the row this item staged a moment ago: an L2 hit), in flight during the second product
    const __amdgpu_buffer_rsrc_t rs_v = rows_rsrc(v + row, 2L * N);
    const int kl2 = 128 * (opaque(lane) >> 5) + (opaque(lane) & 31);
    u16 vold[16];
#pragma unroll
    for (int i = 0; i < 16; i++) vold[i] = (u16)__builtin_amdgcn_raw_buffer_load_b16(rs_v, 2 * kl2, 2 * 32 * ((i & 3) + 8 * (i >> 2)), 0);
    STAMP(4);                                              // e, v's reversed array, e's image and rows
    pi_product_cyc<1>(F, nat, T, g, lane, keep, C);        // e v
    STAMP(5);
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int ko = 32 * ((i & 3) + 8 * (i >> 2));
      u32 nv;                                              // vold - 2^kb (e v): of e v only its residue modulo 2^(m - kb) reaches the low m bits
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(nv) : "v"(C[0][i]), "s"(0 - (int)(1u << kb)), "v"(vold[i]));   // |e v| < 2^23; vold's upper half: masked below
      __builtin_amdgcn_raw_buffer_store_b16((u16)(nv & (mr - 1)), rs_v, 2 * kl2, 2 * ko, 0);
    }
    wave_lds_fence();
    STAMP(6);                                              // the lift's stores issued
  }
}

// Generic per-item product on the matrix cores: both operands < q <= 8192 (multiplyPolynomials + dividePolynomials by I,
// index.js:319-401, with q a power of two).  With a = a0 + 128 a1 and b = b0 + 128 b1 the product is
// a0 b0 + 128 (a0 b1 + a1 b0) + 16384 a1 b1, and 16384 = 0 mod q: three plane products, two accumulator groups, two reversed arrays
// (the digit planes of b) per item, the rows of a0 and a1 in registers.
// ONE (q <= 256): one digit plane per operand.
static __host__ __device__ inline size_t pi_reg_wave_bytes2(const PGeom &g) { return pi_reg_wave_bytes(g) + (size_t)16 * g.tpitch; }

template <bool ONE>
__global__ __launch_bounds__(64 * PI_WAVES) __attribute__((amdgpu_waves_per_eu(ONE ? 4 : 3, 4))) void k_polymul_m(
    PGeom g, u32 q, const u16 *__restrict__ a, const u16 *__restrict__ b, long B, u16 *__restrict__ quot, u16 *__restrict__ rem) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char *nat = lds + (size_t)wave * (ONE ? pi_reg_wave_bytes(g) : pi_reg_wave_bytes2(g));
  u32 *T0 = (u32 *)(nat + pi_nat_bytes(g)), *T1 = T0 + 4 * g.tpitch;
  const int N = g.N, NT = g.NT;
  const bool want_q = quot != nullptr;
  auto chunk_of = [](int ln) { return 2 * (ln & 31) + (ln >> 5); };
  const long item_step = (long)gridDim.x * PI_WAVES;         // the NEXT item's operands are requested early: see k_product_tern_m
  RawChunks<2> rwa, rwb;
  auto request = [&](long it) {
    const long rw = it * N;
    const int ch = chunk_of(opaque(lane));
    const AlignedSrc sa = aligned_src(a + rw, 2L * N), sb = aligned_src(b + rw, 2L * N);
    rwa = load_raw<2>(sa, sa.a0 + 32 * ch, 0);
    rwb = load_raw<2>(sb, sb.a0 + 32 * ch, 0);
  };
  if ((long)blockIdx.x * PI_WAVES + wave < B) request((long)blockIdx.x * PI_WAVES + wave);
  [[maybe_unused]] int stamp_iter = -1;                    // -DNTRU_STAMPS: phase stamps of the first items (tools/phase_stamps_peritem.py)
  for (long item = (long)blockIdx.x * PI_WAVES + wave; item < B; item += item_step) {
    const long row = item * N;
    stamp_iter++;
    STAMP(0);
    v4i a0, a1;
    {
      const int ch = chunk_of(opaque(lane));
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
      v4i b0, b1;
      pi_digits(xa, q, 1u, 16 * ch, N, a0, a1);
      pi_digits(xb, q, 1u, 16 * ch, N, b0, b1);
      STAMP(2);                                            // digits
      pi_build_array_ch(nat, T0, g, lane, ch, b0);
      if (!ONE) pi_build_array_ch(nat, T1, g, lane, ch, b1);
      STAMP(3);                                            // reversed arrays
    }
    v16i XL[2], XH[2];                                     // group 0: a0 b0; group 1: a0 b1 + a1 b0
    if (ONE) {
      v4i F[1] = {a0};
      v16i L[1], H[1];
      pi_product_reg<1>(F, T0, g, lane, L, H);
      XL[0] = L[0]; XH[0] = H[0];
#pragma unroll
      for (int i = 0; i < 16; i++) { XL[1][i] = 0; XH[1][i] = 0; }
    } else {
      // as pi_product_reg, with two fragment streams (the planes of b) and three matrix instructions per half trip
      const int ln = opaque(lane);
      const int y0 = 32 * NT - 1 - (ln & 31) + 16 * (ln >> 5);
      const u32 *tb = T0 + (y0 & 3) * g.tpitch + (y0 >> 2);
      const int tstep = 4 * g.tpitch;                      // dwords from a fragment of b0 to the same fragment of b1
      int seam_up = ln == 32 ? 0 : -1, seam_dn = ln == 31 ? 0 : -1;
      asm volatile("" : "+v"(seam_up), "+v"(seam_dn));
      struct Fr { v4i w0, w1; };
      auto frag = [&](int d) {
        d = d > NT - 1 ? NT - 1 : (d < 1 - NT ? 1 - NT : d);
        const u32 *p = tb - 8 * d, *p1 = p + tstep;
        Fr fr;
        fr.w0 = (v4i){(int)p[0], (int)p[1], (int)p[2], (int)p[3]};
        fr.w1 = (v4i){(int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
        return fr;
      };
      const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const Fr f0 = frag(0);
      Fr la = frag(1), ha = frag(-1), lb, hb;
      {
        u32 mlow[4], mhigh[4];
        pi_diag_low_mask(ln, mlow);
#pragma unroll
        for (int c = 0; c < 4; c++) mhigh[c] = ~mlow[c];
        XL[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, and4(f0.w0, mlow), zero, 0, 0, 0);
        XL[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, and4(f0.w1, mlow), zero, 0, 0, 0);
        XL[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, and4(f0.w0, mlow), XL[1], 0, 0, 0);
        XH[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, and4(f0.w0, mhigh), zero, 0, 0, 0);
        XH[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, and4(f0.w1, mhigh), zero, 0, 0, 0);
        XH[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, and4(f0.w0, mhigh), XH[1], 0, 0, 0);
      }
      v4i AL0 = a0, AL1 = a1, AH0 = a0, AH1 = a1;
      auto trip = [&](const Fr &wl, const Fr &wh) {
        AL0 = rows_up(AL0, seam_up);
        XL[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AL0, wl.w0, XL[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        AL1 = rows_up(AL1, seam_up);
        XL[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AL0, wl.w1, XL[1], 0, 0, 0);
        XL[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AL1, wl.w0, XL[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        AH0 = rows_down(AH0, seam_dn);
        XH[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AH0, wh.w0, XH[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        AH1 = rows_down(AH1, seam_dn);
        XH[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AH0, wh.w1, XH[1], 0, 0, 0);
        XH[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AH1, wh.w0, XH[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      int j = 1;
      for (; j + 1 < NT; j += 2) {
        lb = frag(j + 1); hb = frag(-(j + 1));
        trip(la, ha);
        la = frag(j + 2); ha = frag(-(j + 2));
        trip(lb, hb);
      }
      if (j < NT) trip(la, ha);
    }
    STAMP(4);                                              // matrix loops
    {
      const int ln = opaque(lane), kl = 128 * (ln >> 5) + (ln & 31);     // see k_verify_keys_m: indices >= N are dropped
      const __amdgpu_buffer_rsrc_t rs_r = rows_rsrc(rem + row, 2L * N);
      const __amdgpu_buffer_rsrc_t rs_q = rows_rsrc(want_q ? quot + row : nullptr, want_q ? 2L * N : 0L);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int ko = 32 * ((i & 3) + 8 * (i >> 2));
        const u32 lo = (u32)XL[0][i] + 128u * (u32)XL[1][i], hi = (u32)XH[0][i] + 128u * (u32)XH[1][i];
        __builtin_amdgcn_raw_buffer_store_b16((u16)((lo + hi) & (q - 1)), rs_r, 2 * kl, 2 * ko, 0);
        if (want_q) __builtin_amdgcn_raw_buffer_store_b16((u16)((0u - hi) & (q - 1)), rs_q, 2 * kl, 2 * ko, 0);
      }
    }
    wave_lds_fence();
    STAMP(5);                                              // result stores issued
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
  if (eng->max_blocks_per_cu && eng->max_blocks_per_cu < per_cu) per_cu = eng->max_blocks_per_cu;     // NTRU_MAX_BLOCKS_PER_CU (experiments)
  long blocks = (long)eng->cus * (per_cu < 1 ? 1 : per_cu), work = (B + PI_WAVES - 1) / PI_WAVES;
  if (blocks > work) blocks = work;
  *grid = dim3((unsigned)blocks);
  return NTRU_OK;
}

int ntru_launch_polymul_matrix(ntru_engine *eng, int N, int mod, const uint16_t *d_a, const uint16_t *d_b, int64_t B, uint16_t *d_quot,
                               uint16_t *d_rem) {
  if (!peritem_applies(eng, N, mod)) return NTRU_NOT_TAKEN;
  const PGeom pg = make_pgeom(N);
  dim3 grid;
  auto go = [&](auto kern, size_t lds) -> int {
    if (int rc = peritem_grid(eng, kern, lds, (long)B, &grid)) return rc;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_polymul_m");
    hipLaunchKernelGGL(kern, grid, dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)mod, d_a, d_b, (long)B, d_quot, d_rem);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  };
  return mod <= 256 ? go(k_polymul_m<true>, PI_WAVES * pi_reg_wave_bytes(pg)) : go(k_polymul_m<false>, PI_WAVES * pi_reg_wave_bytes2(pg));
}

bool ntru_product_tern_matrix_applies(const ntru_engine *eng, int N, int q) { return peritem_applies(eng, N, q); }

int ntru_launch_product_tern_matrix(ntru_engine *eng, int N, int q, uint32_t mul, const uint16_t *d_a, const int8_t *d_s, long B,
                                    uint16_t *d_rem) {
  const PGeom pg = make_pgeom(N);
  dim3 grid;
  auto go = [&](auto kern, size_t lds) -> int {
    if (int rc = peritem_grid(eng, kern, lds, B, &grid)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)q, (u32)mul, d_a, d_s, B, d_rem);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  };
  return q <= 256 ? go(k_product_tern_m<true>, PI_WAVES * pi_reg_wave_bytes(pg)) : go(k_product_tern_m<false>, PI_WAVES * pi_reg_wave_bytes(pg));
}

int ntru_launch_verify_keys_matrix(ntru_engine *eng, int N, int q, int p, const int8_t *d_f, const int8_t *d_g, const uint16_t *d_fq,
                                   const uint8_t *d_fp, const uint16_t *d_h, int64_t B, uint16_t *d_quot_fq, uint16_t *d_rem_fq,
                                   uint8_t *d_quot_fp, uint8_t *d_rem_fp, uint16_t *d_quot_h, uint16_t *d_rem_h, uint8_t *d_flags) {
  if (p != 3 || !peritem_applies(eng, N, q)) return NTRU_NOT_TAKEN;
  const PGeom pg = make_pgeom(N);
#ifdef NTRU_EXPERIMENTS
  if (eng->path == 12) {                                   // the 16-row tile (measured slower)
    const size_t lds16 = PI_WAVES * pi_verify16_wave_bytes(pg);
    dim3 grid16;
    if (int rc = peritem_grid(eng, k_verify_keys_m16, lds16, (long)B, &grid16)) return rc;
    snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_verify_keys_m16");
    hipLaunchKernelGGL(k_verify_keys_m16, grid16, dim3(64 * PI_WAVES), lds16, eng->stream, pg, (u32)q, d_f, d_g, d_fq, d_fp, d_h, (long)B,
                       d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp, d_quot_h, d_rem_h, d_flags);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
#endif
  const size_t lds = PI_WAVES * pi_verify_wave_bytes(pg);
  dim3 grid;
  if (int rc = peritem_grid(eng, k_verify_keys_m, lds, (long)B, &grid)) return rc;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_verify_keys_m");
  hipLaunchKernelGGL(k_verify_keys_m, grid, dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)q, d_f, d_g, d_fq, d_fp, d_h, (long)B,
                     d_quot_fq, d_rem_fq, d_quot_fp, d_rem_fp, d_quot_h, d_rem_h, d_flags);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}


// One Newton round (v from kb to m bits) of the key inversion as ONE kernel: kb <= 7 (v below 128: one digit plane), per-item matrix path.
bool ntru_newton_round_matrix_applies(const ntru_engine *eng, int N, int kb, int m) {
  return kb <= 7 && m <= 2 * kb && m > kb && peritem_applies(eng, N, 1 << m);
}
int ntru_launch_newton_round_matrix(ntru_engine *eng, int N, int kb, int m, const int8_t *d_f, uint16_t *d_v, long B) {
  if (!ntru_newton_round_matrix_applies(eng, N, kb, m)) return NTRU_NOT_TAKEN;
  const PGeom pg = make_pgeom(N);
  const size_t lds = PI_WAVES * pi_reg_wave_bytes(pg);
  dim3 grid;
  if (int rc = peritem_grid(eng, k_newton_round_m, lds, B, &grid)) return rc;
  hipLaunchKernelGGL(k_newton_round_m, grid, dim3(64 * PI_WAVES), lds, eng->stream, pg, (u32)kb, (u32)m, d_f, d_v, B);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

#ifdef NTRU_STAMPS
// (diagnostic build only: one Newton round as its own call, for tools/phase_stamps_peritem.py)
extern "C" int ntru_debug_newton_round(ntru_engine *eng, int N, int kb, int m, const int8_t *d_f, uint16_t *d_v, long B) {
  return ntru_launch_newton_round_matrix(eng, N, kb, m, d_f, d_v, B);
}
#endif
