// keygen_sampler_pack.hip -- MI355X (gfx950): the rows SURVEY.md 8(f) widens into, each with its *_dev entry point:
//   key inversion (loadPrivateKeyF / polyInv, index.js:30-49, :491-514): k_invert_key + Newton rounds on the per-item product kernels
//   on-device ternary sampler (generateCustomArray, index.js:461-488) on a ChaCha20 stream: k_sample_ternary
//   BN254 field packing (packOutput / unpackInput, index.js:572-620): k_pack, k_unpack
//   elementwise: stand-alone dividePolynomials(., I, mod) and addPolynomials on ciphertexts (index.js:358-401, :235-244)
#include "kernels_common.h"

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


// ---- on-device ternary sampler: generateCustomArray (index.js:461-488) for one item per LANE ---------------------
// Same procedure as the reference: [1]*n1 ++ [other]*n2 ++ [0]*..., then for i = N-1 .. 1: j = u32 % (i+1), swap.
// The u32 of step t of item b is word t of the ChaCha20 keystream (RFC 8439 block function) under the caller's key with
// nonce (b_lo, b_hi, "NTRU"), so any host can replay it with a stock ChaCha20.  The Fisher-Yates chain is inherently
// sequential per item, so items are spread over lanes.  A lane's row lives in LDS as 2-bit symbols (0, 1, 2 = `other`),
// 16 per dword, laid out [word][lane]: whatever word a lane touches, it is in the lane's own bank.  13 KB per wave at
// N = 821 -> 12 waves per CU, and the launcher asks for as much LDS as makes the resident count a MULTIPLE OF FOUR: the
// kernel is bound by vector issue, every workgroup is one wave, and 9 waves on 4 SIMDs (what the round-2 layout with its
// per-wave reciprocal table got) left three SIMDs idle a third of the time (profiles/archive/r03_ablation_sampler.txt).
// i is wave-uniform: the word that holds position i stays in a register until i leaves it, u32 % (i+1) is a multiply by
// a reciprocal floor(2^32 / d) that arrives through the scalar cache (constant table, d < 2048; an LDS table above),
// one 24-bit multiply-subtract (the remainder is below 2d < 2^24, so the product is only needed modulo 2^24) and one
// correction -- instead of a 35-instruction division.
struct ChaChaKey { u32 k[8]; };
// A workgroup is WAVES = 4 independent waves (no barrier, private LDS regions) wherever four rows regions fit the LDS: as TWELVE
// single-wave workgroups per CU the same kernel took 3.4 ms per 2^20 items at N = 821, as three four-wave workgroups 2.2 ms
// (profiles/archive/r03_ab_sampler.txt; the inversion kernel, eight single-wave workgroups per CU, does not care: more than eight workgroups
// per CU do not seem to be resident together, whatever the occupancy query says).
struct RecipTable {
  u32 v[2048];
  constexpr RecipTable() : v() { for (unsigned d = 2; d < 2048; d++) v[d] = (u32)(0x100000000ULL / d); }
};
static __constant__ const RecipTable g_recip = RecipTable();

#define CHACHA_QR(a, b, c, d)                                                          \
  a += b; d ^= a; d = __builtin_rotateleft32(d, 16); c += d; b ^= c; b = __builtin_rotateleft32(b, 12); \
  a += b; d ^= a; d = __builtin_rotateleft32(d, 8);  c += d; b ^= c; b = __builtin_rotateleft32(b, 7);

// word w (symbols 16 w .. 16 w + 15) of the start row [1]*n1 ++ [2]*n2 ++ [0]*...
static __device__ __forceinline__ u32 sampler_start_word(int w, int n1, int n2) {
  auto below = [](int k) { return k <= 0 ? 0u : (k >= 16 ? 0xFFFFFFFFu : (1u << (2 * k)) - 1u); };   // symbols 0 .. k-1 of a word
  const u32 m1 = below(n1 - 16 * w), m2 = below(n1 + n2 - 16 * w);
  return (0x55555555u & m1) | (0xAAAAAAAAu & m2 & ~m1);
}

// DR: double rounds of the block function: 10 = ChaCha20 (RFC 8439, the default), 6 = ChaCha12, 4 = ChaCha8 (ntru_engine_set_sampler_rounds:
// the kernel is bound by the vector issue of these rounds; generateCustomArray's contract is the shuffle and its `u32 % (i + 1)` draws,
// not the generator behind them, and a host replays whichever variant it asked for).
template <bool BIGN, int WAVES, int DR>     // BIGN: N + 1 >= 2048, reciprocals from an LDS table behind the rows
__global__ __launch_bounds__(WAVES * 64) void k_sample_ternary(int N, int n1, int n2, u32 other, ChaChaKey key,
                                                       unsigned long long first_item, long B,
                                                       uint8_t *__restrict__ out, int NW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t per_wave = (size_t)NW * 64 + (BIGN ? (size_t)((N + 2) & ~1) : 0);     // dwords
  u32 *rows = (u32 *)lds + wave * per_wave;              // [NW][64] dwords of 16 symbols
  u32 *recip_l = rows + (size_t)NW * 64;                 // BIGN: [N + 1]
  if (BIGN) {
    for (int d = lane; d <= N; d += 64) recip_l[d] = d >= 2 ? (u32)(0x100000000ULL / (unsigned)d) : 0u;
    wave_lds_fence();
  }
  auto recip_of = [&](int d) -> u32 { return d < 2 ? 0u : (BIGN ? recip_l[d] : g_recip.v[d]); };
  u32 *col = rows + lane;                                // word w of this lane's row: col[64 w]
  const bool out_aligned = (((unsigned long long)out) & 15) == 0 && N >= 16;
  for (long base = ((long)blockIdx.x * WAVES + wave) * 64; base < B; base += (long)gridDim.x * WAVES * 64) {
    for (int w = 0; w < NW; w++) col[64 * w] = sampler_start_word(w, n1, n2);
    const unsigned long long item = first_item + (unsigned long long)(base + lane);
    const u32 n0 = (u32)item, nn1 = (u32)(item >> 32), nn2 = 0x4e545255u;
    int i = N - 1;
    u32 a = col[64 * (i >> 4)];                          // the word that holds position i
    for (u32 ctr = 0; i >= 1; ctr++) {                  // i is the same in every lane: uniform loop
      u32 rc[16];
      if (!BIGN) {                                       // requested now, needed after the rounds
#pragma unroll
        for (int w = 0; w < 16; w++) rc[w] = recip_of(i + 1 - w);
      }
      u32 x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
      u32 x4 = key.k[0], x5 = key.k[1], x6 = key.k[2], x7 = key.k[3], x8 = key.k[4], x9 = key.k[5], x10 = key.k[6],
          x11 = key.k[7], x12 = ctr, x13 = n0, x14 = nn1, x15 = nn2;
#pragma unroll
      for (int r = 0; r < DR; r++) {
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
          const u32 hi = __umulhi(ks[w], BIGN ? recip_of((int)d) : rc[w]);
          u32 j;                                                         // ks - hi d is in [0, 2d): its low 24 bits are all of it
          asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(j) : "v"(hi), "s"(0 - (int)d), "v"(ks[w]));
          j &= 0xFFFFFFu;
          j = min(j, j - d);
          const int wi = i >> 4, si = 2 * (i & 15), wj = (int)(j >> 4), sj = 2 * (int)(j & 15);
          const u32 bl = col[64 * wj];
          const bool same = wj == wi;
          const u32 b = same ? a : bl;
          const u32 x = ((a >> si) ^ (b >> sj)) & 3u;                    // swap two 2-bit fields by their difference
          a ^= x << si;
          const u32 nb = (same ? a : b) ^ (x << sj);
          col[64 * wj] = nb;
          a = same ? nb : a;
          i--;
          if ((i & 15) == 15) {                                          // uniform: position i has moved into the word below
            col[64 * wi] = a;
            a = col[64 * (i >> 4)];
          }
        }
      }
    }
    col[0] = a;                                          // i == 0: the register copy of word 0 (N == 1: unchanged)
    wave_lds_fence();
    if (out_aligned && base + 64 <= B) {
      // rows -> row-major byte output.  The 64 rows of the block are ONE contiguous run of 64 N bytes that starts on a 64-byte
      // boundary: lane L writes the 16-byte pieces of bytes [S L, S L + S), S = 16 NW >= N, i.e. the end of row ~L and the start of
      // row ~L + 1 (distinct rows per lane: distinct LDS banks), every store aligned; 64 N is a multiple of 16, so a piece is
      // inside the run or outside it.
      const u32 S = 16u * (u32)NW, a0 = S * (u32)lane, run = 64u * (u32)N;
      u32 r = a0 / (u32)N, c = a0 - r * (u32)N;
      uint4 *dst = (uint4 *)(out + (size_t)base * N) + (size_t)NW * lane;
      for (int it = 0; it < NW; it++) {
        if (a0 + 16u * (u32)it < run) {
          const u32 *rp = rows + r;
          const int w = (int)(c >> 4), w1i = w + 1 < NW ? w + 1 : w;
          u32 bits = __builtin_amdgcn_alignbit(rp[64 * w1i], rp[64 * w], 2 * (c & 15));
          const int nf = N - (int)c;                     // symbols of row r from c on (>= 1)
          if (nf < 16) bits = (bits & ((1u << (2 * nf)) - 1u)) | (rows[r + 1] << (2 * nf));   // the piece continues in row r + 1
          u32 o[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            u32 t = (bits >> (8 * q)) & 0xFFu;
            t = (t | (t << 12)) & 0x000F000Fu;
            t = (t | (t << 6)) & 0x03030303u;
            if (other != 2u) t += ((t >> 1) & 0x01010101u) * (other - 2u);
            o[q] = t;
          }
          dst[it] = make_uint4(o[0], o[1], o[2], o[3]);
        }
        c += 16;
        if (c >= (u32)N) { c -= (u32)N; r++; }
      }
    } else {
      // any alignment, partial last block: the wave walks one row at a time, one byte per lane
      for (int rr = 0; rr < 64; rr++) {
        if (base + rr >= B) break;
        uint8_t *dst = out + (size_t)(base + rr) * N;
        const u32 *src = rows + rr;
        for (int k = lane; k < N; k += 64) {
          const u32 sym = (src[64 * (k >> 4)] >> (2 * (k & 15))) & 3u;
          dst[k] = (uint8_t)(sym == 2u ? other : sym);
        }
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
// planes in LDS, [array][word][lane]; GF(3) uses two planes per polynomial: plane 0 = coefficient is non-zero, plane 1 = its sign (set: the
// coefficient is 2 = -1; whatever it holds where plane 0 is clear is never looked at).

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
    // gg = rev_{N-1}(f) as bit planes: plane bit j of a key is the residue flag of f[N-1-j].  A whole block of 64 keys is ONE contiguous,
    // 64-byte aligned run of 64 N bytes: lane L loads its 16-byte pieces 16 (L + 64 it), turns the 16 int8 into residue flags on
    // packed bytes, and ORs the (reversed) bit field into the words of the key(s) the piece belongs to, in LDS as [plane][word][key];
    // then every lane reads its own key's words.  (A byte load per coefficient from every lane's own row -- 64 lines per
    // instruction -- was a fifth of the mod-2 inversion.)
    const bool whole_in = k0 + 64 <= B && N >= 32 && (((unsigned long long)f & 15) == 0);
    u32 *tin = NWC ? base : base + (size_t)(AG * PL) * NW * 64;
    if (whole_in) {
      for (int w = 0; w < PL * NW; w++) tin[w * 64 + lane] = 0;
      wave_lds_fence();
      const uint4 *src = (const uint4 *)(f + (size_t)k0 * N);
      const u32 uN = (u32)N, inv = 0xFFFFFFFFu / uN + 1u, total = 64u * uN;       // e / N = umulhi(e, inv): exact while 64 N^2 < 2^32
      auto put = [&](u32 *pw, u32 lo, u32 n, u32 x) {       // coefficients c .. c + n - 1 (bit j of x: c + j) = plane bits lo + n - 1 .. lo of one key
        const u32 rev = __builtin_bitreverse32(x & ((1u << n) - 1u)) >> (32u - n);
        const u32 w = lo >> 5, sh = lo & 31u;
        if (rev) {
          atomicOr(pw + 64 * w, rev << sh);
          if (sh + n > 32u && (rev >> (32u - sh))) atomicOr(pw + 64 * (w + 1), rev >> (32u - sh));
        }
      };
      for (u32 e = 16u * (u32)lane, it = 0; e < total; e += 1024u, it++) {
        const uint4 d4 = src[lane + 64 * it];
        const u32 d[4] = {d4.x, d4.y, d4.z, d4.w};
        u32 x0 = 0, x1 = 0;                                 // bit j: byte j is 1 / is 2 modulo P
#pragma unroll
        for (int c = 0; c < 4; c++) {
          u32 r0, r1 = 0;
          if (P == 2) {
            r0 = d[c] & 0x01010101u;                        // the parity of a two's complement byte is its residue modulo 2
          } else {
            // int8 modulo 3 on four packed bytes: 4 = 1 (mod 3), so a byte is congruent to the sum of its base-4 digits, + 2 if its sign
            // bit is set (-256 = 2 mod 3); two more digit sums bring that into 0..3, where 3 stands for 0
            const u32 s1 = (d[c] & 0x33333333u) + ((d[c] >> 2) & 0x33333333u);
            const u32 s2 = (s1 & 0x0F0F0F0Fu) + ((s1 >> 4) & 0x0F0F0F0Fu) + ((d[c] >> 6) & 0x02020202u);
            const u32 u = (s2 & 0x03030303u) + ((s2 >> 2) & 0x03030303u);
            const u32 v = (u & 0x03030303u) + ((u >> 2) & 0x03030303u);
            r0 = v & ~(v >> 1) & 0x01010101u;
            r1 = (v >> 1) & ~v & 0x01010101u;
          }
          x0 |= (((r0 * 0x00204081u) >> 21) & 15u) << (4 * c);          // bit 0 of byte j -> bit j
          x1 |= (((r1 * 0x00204081u) >> 21) & 15u) << (4 * c);
        }
        const u32 kk = __umulhi(e, inv), i = e - kk * uN, nf = uN - i, na = nf < 16u ? nf : 16u;
        put(tin + kk, nf - na, na, x0);
        if (P == 3) put(tin + (size_t)NW * 64 + kk, nf - na, na, x1);
        if (nf < 16u) {                                     // the piece goes on in row kk + 1 (which exists: 64 N is a multiple of 16)
          put(tin + kk + 1, uN - (16u - nf), 16u - nf, x0 >> nf);
          if (P == 3) put(tin + (size_t)NW * 64 + kk + 1, uN - (16u - nf), 16u - nf, x1 >> nf);
        }
      }
      wave_lds_fence();
    }
    // ff = rev(x^N - 1) = 1 - x^N, gg, vv = 0, ww = 1
#pragma unroll UNR
    for (int w = 0; w < NW; w++) {
      u32 g1 = 0, g2 = 0;
      if (whole_in) {
        g1 = tin[(0 * NW + w) * 64 + lane];
        if (P == 3) g2 = tin[(1 * NW + w) * 64 + lane];
      } else
#pragma unroll 1                                           // (fully unrolled, the 32 NW tests `i < N` are hoisted out of the key loop and
      for (int b0 = 0; b0 < 32; b0 += 8) {                //  spilled as ~1650 scalar registers; eight byte loads in flight at a time)
#pragma unroll
        for (int bb = 0; bb < 8; bb++) {
          const int b = b0 + bb, i = 32 * w + b;          // coefficient i of gg is f[N-1-i]
          if (i < N) {                                    // (lanes without a key work on key 0's row; nothing of theirs is stored)
            // residue of the int8 in [0, P) without compares: 129 = 0 mod 3, and the parity of a two's complement byte is its residue mod 2
            const u32 c = P == 2 ? (u32)fk[N - 1 - i] & 1u : (u32)(fk[N - 1 - i] + 129) % 3u;
            g1 |= (c & 1u) << b;
            g2 |= (c >> 1) << b;
          }
        }
      }
      const u32 top = (32 * w <= N && N < 32 * w + 32) ? 1u << (N & 31) : 0u;   // coefficient N of ff is -1
      if (P == 2) {
        at(AF, 0, w) = (w == 0 ? 1u : 0u) | top; at(AG, 0, w) = g1;
        at(AV, 0, w) = 0; at(AW, 0, w) = w == 0 ? 1u : 0u;
      } else {                                            // (non-zero, sign): 1 - x^N = (1, +), (x^N, -); g1 / g2 = residue is 1 / is 2
        at(AF, 0, w) = (w == 0 ? 1u : 0u) | top; at(AF, 1, w) = top;
        at(AG, 0, w) = g1 | g2; at(AG, 1, w) = g2;
        at(AV, 0, w) = 0; at(AV, 1, w) = 0; at(AW, 0, w) = w == 0 ? 1u : 0u; at(AW, 1, w) = 0;
      }
    }
    int delta = 1;
    auto wave_max_abs_delta = [&]() {
      const u32 ad = have ? (u32)(delta < 0 ? -delta : delta) : 0u;          // (lanes without a key work on key 0's row)
      u32 mx = 0;
      for (int bit = 16; bit >= 0; bit--) {                 // wave maximum by bisection on ballots: scalar work only
        const u32 c = mx | (1u << bit);
        if (__ballot(ad >= c) != 0) mx = c;
      }
      return (int)mx;
    };
    auto word_of = [&](int deg) { const int t = deg >> 5; return t < NW - 1 ? t : NW - 1; };
    int fg_top = NW - 1, vw_top = 0;
    // GF(3) on (non-zero, sign) plane pairs: r = g + cm f with the per-lane scalar cm given as two lane masks (m2: cm == 2, mnz: cm != 0).
    // Six three-input functions (v_bitop3_b32 / v_bfi_b32 each; eight with planes "is 1" / "is 2"): t = cm f is (f_n & mnz, f_s ^ m2)
    // -- the scalar 2 flips the sign --; the sum is non-zero iff exactly one term is, or both are and their signs agree (1 + 1 = 2:
    // then the sign flips); its sign is g's (flipped if t is non-zero too) where g is non-zero, t's otherwise.  Sign bits under a clear
    // non-zero bit are garbage and never reach a result that is used (tests/test_divstep_bounds_cpu.py checks the formulas exhaustively).
    auto madd3 = [](u32 g0, u32 g1, u32 f0, u32 f1, u32 m2, u32 mnz, u32 &r0, u32 &r1) {
      // (v_bitop3_b32 by hand -- truth table = the expression on a = 0xf0, b = 0xcc, c = 0xaa: left to itself the compiler splits r0 in two)
      const u32 tn = f0 & mnz, ts = f1 ^ m2;
      const u32 x = __builtin_amdgcn_bitop3_b32(g1, f1, m2, 0x96);          // g1 ^ ts
      r0 = __builtin_amdgcn_bitop3_b32(g0, tn, x, 0x7c);                   // (g0 ^ tn) | (g0 & tn & ~x)
      const u32 p = g1 ^ tn;
      r1 = __builtin_amdgcn_bitop3_b32(g0, p, ts, 0xca);                   // g0 ? p : ts
    };
    for (int step = 0; step < 2 * N - 1; step++) {
      if ((step & 15) == 0) {                             // (before delta moves on: the bounds are for what this step reads)
        const int mx = wave_max_abs_delta();
        fg_top = word_of((2 * N - 1 - step + mx) >> 1);
        vw_top = word_of((step + mx + 34) >> 1);
      }
      const u32 f0w0 = at(AF, 0, 0), g0w0 = at(AG, 0, 0);
      const u32 f0w1 = P == 3 ? at(AF, 1, 0) : 0u, g0w1 = P == 3 ? at(AG, 1, 0) : 0u;
      // constant terms (GF(3): non-zero ? 1 + sign : 0)
      const int fc = P == 3 ? ((f0w0 & 1u) ? 1 + (int)(f0w1 & 1u) : 0) : (int)(f0w0 & 1u);
      const int gc = P == 3 ? ((g0w0 & 1u) ? 1 + (int)(g0w1 & 1u) : 0) : (int)(g0w0 & 1u);
      const bool swap = delta > 0 && gc != 0;
      delta = (swap ? -delta : delta) + 1;
      const int c1 = swap ? gc : fc;                      // new f(0): multiplies g and w
      const int c2 = (P - (swap ? fc : gc)) % P;           // -(new g(0)): multiplies f and v
      const u32 c2m1 = c2 == 1 ? ~0u : 0u;
      // GF(3): g and w are scaled by the unit 1 / f(0) every step (they stay consistent with each other, and the inverse
      // is unique), so ONE scalar multiplies f and v: new g = (g - (g(0) / f(0)) f) / x, and 1 / f(0) = f(0) in GF(3)
      const int cm = P == 3 ? (9 - (swap ? fc : gc) * c1) % 3 : 0;
      u32 m2 = cm == 2 ? ~0u : 0u, mnz = cm != 0 ? ~0u : 0u;
      asm volatile("" : "+v"(m2), "+v"(mnz));            // opaque: `x & mask` must stay a bit operation that fuses, not become a select
      // Degrees (by induction over the two kinds of step; delta = 1, deg f = N, deg g < N, v = 0, w = 1 at the start), before step n:
      //   2 deg f <= 2N - 1 - n + delta,  2 deg g <= 2N - 1 - n - delta        (f and g shrink),
      //   2 deg v <= n - 1 + delta,       2 deg w <= n + 1 - delta             (v and w grow half as fast as the step counter),
      // so a step touches about N / 32 + 2 words of the four polynomials together instead of 4 N / 32.  delta is per lane: the
      // maximum of |delta| over the wave's keys is taken every 16th step (at the top of the step).  For f and g the bound at that step
      // holds for the 15 that follow (n - |delta| never decreases); for v and w those steps are allowed for (|delta| and n move by at
      // most one per step, and a step writes the polynomials of step n + 1: + 34 in all).  The tests are scalar branches per word.
      // Words from the TOP down: g / x takes its incoming bit from the word above (already computed: `gup`), x v from the word below
      // (not yet overwritten), so every word is read and rewritten in place within its own iteration -- with register planes nothing
      // has to be copied at the end of a step (going up, the results land one word behind and the loop ended in ~100 moves).
      u32 gup[PL];
#pragma unroll
      for (int pl = 0; pl < PL; pl++) gup[pl] = 0;
      auto fg_word = [&](int w) {
        u32 F[PL], G[PL];
#pragma unroll
        for (int pl = 0; pl < PL; pl++) {
          F[pl] = at(AF, pl, w); G[pl] = at(AG, pl, w);
          const u32 f_ = F[pl];                                             // conditional swap as selects: the condition is per lane, not per bit
          F[pl] = swap ? G[pl] : f_; G[pl] = swap ? f_ : G[pl];
        }
        u32 NG[PL];
        if (P == 2) {                                     // c1 = 1; c2 = g(0)
          NG[0] = G[0] ^ (c2m1 & F[0]);
        } else {
          madd3(G[0], G[1], F[0], F[1], m2, mnz, NG[0], NG[1]);
        }
#pragma unroll
        for (int pl = 0; pl < PL; pl++) {
          at(AF, pl, w) = F[pl];
          at(AG, pl, w) = __builtin_amdgcn_alignbit(gup[pl], NG[pl], 1);       // g = g / x (gup = 0 above the top word: zero there)
          gup[pl] = NG[pl];
        }
      };
      auto vw_word = [&](int w) {
        u32 V[PL], W[PL], NWW[PL];
#pragma unroll
        for (int pl = 0; pl < PL; pl++) {
          W[pl] = at(AW, pl, w);
          V[pl] = __builtin_amdgcn_alignbit(at(AV, pl, w), w > 0 ? at(AV, pl, w - 1) : 0u, 31);        // v = x v
          const u32 v_ = V[pl];
          V[pl] = swap ? W[pl] : v_; W[pl] = swap ? v_ : W[pl];
        }
        if (P == 2) {
          NWW[0] = W[0] ^ (c2m1 & V[0]);
        } else {
          madd3(W[0], W[1], V[0], V[1], m2, mnz, NWW[0], NWW[1]);
        }
#pragma unroll
        for (int pl = 0; pl < PL; pl++) { at(AV, pl, w) = V[pl]; at(AW, pl, w) = NWW[pl]; }
      };
      // One test per GROUP of words (a group is worked on when its lowest word is inside the bound), marked likely: the bodies then
      // stay in line and a group that is worked on costs a compare and a branch that is not taken.  (Left to itself the compiler
      // moved every body out of line, two taken branches per word; GF(2) has four instructions per word, so small groups cost it
      // more in tests than they save.)
      constexpr int GRP = P == 2 ? 8 : 2;                  // measured at N = 821, 2^18 keys: GF(2) 3.3 / 2.5 / 2.6 / 3.0 ms with 4 / 8 / 16 / all words; GF(3) 6.5 / 6.2 / 6.4 with 1 / 2 / 4
#pragma unroll UNR
      for (int gb = (NW - 1) / GRP * GRP; gb >= 0; gb -= GRP) {
        if (__builtin_expect(gb <= fg_top, 1)) {
#pragma unroll
          for (int w = gb + GRP - 1; w >= gb; w--) if (w < NW) fg_word(w);
        }
        if (__builtin_expect(gb <= vw_top, 1)) {
#pragma unroll
          for (int w = gb + GRP - 1; w >= gb; w--) if (w < NW) vw_word(w);
        }
      }
    }
    // unit iff the gcd (in ff) is a non-zero constant.  Words above the final bound were left alone once they had become zero
    // (they may still hold what they held then), so only the words below it are looked at.
    fg_top = word_of((wave_max_abs_delta()) >> 1);       // n = 2N - 1
    u32 rest = 0;
#pragma unroll UNR
    for (int w = 0; w < NW; w++)
      if (w <= fg_top) rest |= at(AF, 0, w) & (w == 0 ? ~1u : ~0u);            // (plane 0 says "non-zero" for both fields)
    const int fc = (at(AF, 0, 0) & 1u) ? 1 + (P == 3 ? (int)(at(AF, 1, 0) & 1u) : 0) : 0;
    const bool unit = rest == 0 && fc != 0;
    if (have && !unit) flags[key] = (uint8_t)(flags[key] | flag_bit);
    // inverse[i] = fc^-1 * vv[N-1-i]; in GF(3) fc^-1 = fc (the scalar 2 flips the signs); zero for a non-unit
    const bool whole = k0 + 64 <= B && N >= 32 && ((((unsigned long long)out16 | (unsigned long long)out8) & 15) == 0);
    if (whole) {
      // The 64 rows of the block are ONE contiguous run of 64 N results that starts on a 64-element boundary.  A store per
      // coefficient from every lane's own row (64 lines touched per instruction, 821 of them per block) was a THIRD of the kernel's
      // time; the finished planes go through LDS as [plane][word][lane] instead, and lane L then writes the 16-element pieces
      // 16 (L + 64 it) of the run: 16 consecutive coefficients are 16 consecutive plane bits (descending), so a piece is one
      // bit-field read of the key's words (two when it crosses into the next row), reversed and spread to bytes or u16 pairs.
      u32 *tr = NWC ? base : base + (size_t)(AV * PL) * NW * 64;
#pragma unroll UNR
      for (int w = 0; w < NW; w++) {
        u32 p0 = at(AV, 0, w), p1 = 0;                      // -> planes "is 1" / "is 2" of fc^-1 vv
        if (P == 3) {
          const u32 nz = p0, sg = fc == 2 ? ~at(AV, 1, w) : at(AV, 1, w);
          p0 = nz & ~sg; p1 = nz & sg;
        }
        p0 = unit ? p0 : 0u; p1 = unit ? p1 : 0u;
        tr[(0 * NW + w) * 64 + lane] = p0;
        if (P == 3) tr[(1 * NW + w) * 64 + lane] = p1;
      }
      wave_lds_fence();
      const u32 uN = (u32)N, inv = 0xFFFFFFFFu / uN + 1u, total = 64u * uN;       // e / N = umulhi(e, inv): exact while 64 N^2 < 2^32
      auto seg = [&](const u32 *pw, u32 nf) {               // the (up to) 16 coefficients from N - nf on, coefficient order, of one plane of one key
        u32 x;
        if (nf >= 16u) {
          const u32 lo = nf - 16u, w = lo >> 5;
          const u32 hi_w = (int)w + 1 < NW ? pw[64 * (w + 1)] : 0u;
          x = __builtin_amdgcn_alignbit(hi_w, pw[64 * w], lo & 31u) & 0xFFFFu;
        } else {
          x = (pw[0] << (16u - nf)) & 0xFFFFu;
        }
        return __builtin_bitreverse32(x) >> 16;
      };
      auto spread4 = [](u32 nib) { return (nib * 0x00204081u) & 0x01010101u; };       // bit j of a nibble -> byte j
      for (u32 e = 16u * (u32)lane; e < total; e += 1024u) {
        const u32 kk = __umulhi(e, inv), i = e - kk * uN, nf = uN - i;
        u32 f0 = seg(tr + kk, nf), f1 = P == 3 ? seg(tr + (size_t)NW * 64 + kk, nf) : 0u;
        if (nf < 16u) {                                     // the piece goes on in row kk + 1 (which exists: 64 N is a multiple of 16)
          f0 |= seg(tr + kk + 1, uN) << nf;
          if (P == 3) f1 |= seg(tr + (size_t)NW * 64 + kk + 1, uN) << nf;
          f0 &= 0xFFFFu; f1 &= 0xFFFFu;
        }
        const size_t at0 = (size_t)k0 * N + e;
        if (out8) {
          u32 o[4];
#pragma unroll
          for (int c = 0; c < 4; c++) o[c] = spread4((f0 >> (4 * c)) & 15u) + 2u * spread4((f1 >> (4 * c)) & 15u);
          *(uint4 *)(out8 + at0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (out16) {
          u32 o[8];
#pragma unroll
          for (int c = 0; c < 8; c++) {
            const u32 s0 = (f0 >> (2 * c)) & 3u, s1 = (f1 >> (2 * c)) & 3u;
            o[c] = ((s0 & 1u) | (s0 & 2u) << 15) + 2u * ((s1 & 1u) | (s1 & 2u) << 15);
          }
          *(uint4 *)(out16 + at0) = make_uint4(o[0], o[1], o[2], o[3]);
          *(uint4 *)(out16 + at0 + 8) = make_uint4(o[4], o[5], o[6], o[7]);
        }
      }
      wave_lds_fence();                                     // (LDS planes: the next block's set-up writes them)
    } else if (have) {
      // partial last block, unaligned outputs, tiny N: one store per coefficient from every lane's own row
#pragma unroll UNR
      for (int w = 0; w < NW; w++) {
        const u32 p0 = at(AV, 0, w), p1 = P == 3 ? at(AV, 1, w) : 0u;
#pragma unroll 1
        for (int b = 0; b < 32; b++) {
          const int i = N - 1 - (32 * w + b);
          if (i < 0) break;
          int c = (int)((p0 >> b) & 1u);
          if (P == 3) c = c ? 1 + (int)((p1 >> b) & 1u) : 0;
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
__global__ void k_newton_combine(u16 *__restrict__ v, const u16 *__restrict__ u, long first, long n, u32 q) {
  for (long i = first + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    v[i] = (u16)((2u * v[i] - u[i]) & (q - 1));
}
// flags |= extra, bytewise (the mod-p inversion's flag bits, collected on a stream of their own)
__global__ void k_or_bytes(uint8_t *__restrict__ flags, const uint8_t *__restrict__ extra, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    flags[i] = (uint8_t)(flags[i] | extra[i]);
}
__global__ void k_newton_combine_vec(u16x8 *__restrict__ v, const u16x8 *__restrict__ u, long nvec, u32 q) {   // 16 bytes per lane
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x)
    v[i] = ((u16)2 * v[i] - u[i]) & (u16)(q - 1);                        // q | 2^16: wrapped 16-bit arithmetic is exact mod q
}

// ---- BN254 field-element packing (index.js:572-620): elementwise, HBM-bound ---------------------------------------
// One thread per 64-bit limb of the output: out[b][o] = sum_j data[b][o*per + j] << (j*bits), four LE limbs per element.
// T: u16 values (ciphertexts, witness arrays) or bytes (decrypted values, ternary arrays).
template <class T>
__global__ void k_pack(int bits, int per, int data_len, int out_size, const T *__restrict__ data, long B,
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

// packOutput of BYTES with 2-bit fields (values of decryptBits, ternary rows: max_val 2 or 3; 126 values per element): one thread per
// output DWORD = 16 consecutive values = 16 consecutive bytes of the row (the eighth dword of an element holds 14), read as four
// dwords at whatever alignment they have.  (The generic kernel above reads 32 single bytes per 64-bit limb: 1.2 ms per 2^20 rows of
// N = 821 against 0.3 ms.)
__global__ void k_pack_bytes2(int data_len, int out_size, const uint8_t *__restrict__ data, long B, u32 *__restrict__ out) {
  const long total = B * out_size * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long b = idx / (out_size * 8);
    const int dw = (int)(idx - b * out_size * 8), o = dw >> 3, k = dw & 7, i0 = 126 * o + 16 * k;
    const uint8_t *src = data + b * data_len + i0;
    u32 res = 0;
    if (i0 + 16 <= data_len) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        u32 d;
        __builtin_memcpy(&d, src + 4 * c, 4);
        u32 t = d & 0x03030303u;
        t |= t >> 6;
        t = (t | (t >> 12)) & 0xFFu;
        res |= t << (8 * c);
      }
    } else {
      for (int j = 0; j < 16; j++) if (i0 + j < data_len) res |= ((u32)src[j] & 3u) << (2 * j);
    }
    out[idx] = k == 7 ? res & 0x0FFFFFFFu : res;
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

// ---- host side: *_dev entry points ------------------------------------------------------------------------------------------

static int check_elementwise(const ntru_engine *eng, int N, int mod, long B) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0) return fail(NTRU_ERR_ARG, "negative batch size");
  if (N < 1 || mod < 2 || mod > 65536) return fail(NTRU_ERR_UNSUPPORTED, "need N >= 1 and 2 <= mod <= 65536");
  return NTRU_OK;
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
      hipLaunchKernelGGL(k_add_mod_vec<true>, elementwise_grid(eng, nvec, true), dim3(256), 0, eng->stream, (u32)mod,
                         (const u16x8 *)d_a, (const u16x8 *)d_b, nvec, (u16x8 *)d_out);
    else
      hipLaunchKernelGGL(k_add_mod_vec<false>, elementwise_grid(eng, nvec, true), dim3(256), 0, eng->stream, (u32)mod,
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
  // planes in LDS (NWC = 0), or only the transposition buffer of the results: [plane][word][lane]
  const size_t lds = NWC ? (size_t)(P == 2 ? 1 : 2) * NWC * 64 * 4 : (size_t)(P == 2 ? 4 : 8) * ((N + 32) / 32) * 64 * 4;
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
  if (int rc = ntru_check_common(eng, N, q, B)) return rc;
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
  // The mod-p inversion (k_invert_key<3>: bit-sliced division steps, vector-ALU bound, no LDS at the usual sizes) depends on f only, and
  // the chain for fq -- mod-2 inversion, then Newton rounds on the per-item matrix kernels (matrix cores + LDS) -- does not depend on
  // it: the two run SIDE BY SIDE, the mod-p inversion on an engine-owned stream forked from the caller's stream and joined back into it
  // before this call returns.  Its flag bits go to a byte array of their own (two kernels must not read-modify-write the same bytes
  // at the same time) and are OR-ed into d_flags behind the join.
  int k = 0;
  while ((1 << k) < q) k++;
  int rounds = 0;
  while ((1 << rounds) < k) rounds++;
  // precision schedule, halved backwards from log2 q (12: 1, 2, 3, 6, 12; 13: 1, 2, 4, 7, 13): as many rounds as doubling from 1 takes, but
  // the v a round starts from has at most 7 bits whenever log2 q <= 14, i.e. ONE non-negative int8 digit plane in every product
  int bits_of[18];
  bits_of[rounds] = k;
  for (int r = rounds; r > 0; r--) bits_of[r - 1] = (bits_of[r] + 1) / 2;
  const int64_t C = B < INVERT_CHUNK ? B : INVERT_CHUNK;     // Newton temporaries for C keys at a time
  const size_t row = (size_t)N * 2, part = rounds > 0 ? ((size_t)C * row + 255) & ~(size_t)255 : 0;
  const size_t fl_bytes = d_fp ? ((size_t)B + 255) & ~(size_t)255 : 0;
  ScratchHold hold(eng, 4 * part + fl_bytes + 256);          // engine-owned, grown on demand, never per call; released (event
  if (hold.rc) return hold.rc;                               // recorded) on every way out of this function
  char *const sc = hold.p;
  uint8_t *const flags_p = (uint8_t *)(sc + 4 * part);
  hipStream_t main_stream = eng->stream;
  bool forked = false;
  if (d_fp) {
    if (!eng->st_aux) HIP_TRY(hipStreamCreateWithFlags(&eng->st_aux, hipStreamNonBlocking));
    if (!eng->ev_fork) HIP_TRY(hipEventCreateWithFlags(&eng->ev_fork, hipEventDisableTiming));
    if (!eng->ev_join) HIP_TRY(hipEventCreateWithFlags(&eng->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(eng->ev_fork, main_stream));      // f is ready, the scratch buffer is ours (behind the memset above)
    HIP_TRY(hipStreamWaitEvent(eng->st_aux, eng->ev_fork, 0));
    HIP_TRY(hipMemsetAsync(flags_p, 0, (size_t)B, eng->st_aux));
    eng->stream = eng->st_aux;
    const int rc3 = launch_invert<3>(eng, N, d_f, (long)B, nullptr, d_fp, flags_p, NTRU_FLAG_NOT_UNIT_MODP);
    eng->stream = main_stream;
    // (joined below even when a launch fails: what is enqueued on the forked stream must not outlive this call)
    const hipError_t ej = hipEventRecord(eng->ev_join, eng->st_aux);
    forked = ej == hipSuccess;
    // Without a join event nothing orders the caller's stream (or the scratch buffer's next user) behind what is already enqueued on the
    // forked stream -- the mod-p kernel writes d_fp and the flag bytes in the shared scratch: wait for that stream on the host instead.
    if (!forked) (void)hipStreamSynchronize(eng->st_aux);
    if (rc3) { if (forked) (void)hipStreamWaitEvent(main_stream, eng->ev_join, 0); return rc3; }
    HIP_TRY(ej);
  }
  auto join = [&]() -> int {
    if (!forked) return NTRU_OK;
    forked = false;
    HIP_TRY(hipStreamWaitEvent(main_stream, eng->ev_join, 0));
    hipLaunchKernelGGL(k_or_bytes, elementwise_grid(eng, B), dim3(256), 0, main_stream, d_flags, (const uint8_t *)flags_p, (long)B);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  };
  struct Joiner { decltype(join) &j; ~Joiner() { (void)j(); } } joiner{join};      // every early return below joins first
  // mod 2 inverse straight into d_fq (as 0/1 coefficients), then Newton rounds v <- 2v - f v^2 mod q (index.js:499-506;
  // the reference runs log2(q) - 1 of them, the unique inverse mod q is reached once 2^rounds >= log2(q))
  if (int rc = launch_invert<2>(eng, N, d_f, (long)B, d_fq, nullptr, d_flags, NTRU_FLAG_NOT_UNIT_MOD2)) return rc;
  if (rounds > 0) {
    struct { void *p; } f16{sc}, t{sc + part}, u{sc + 2 * part}, qs{sc + 3 * part};
    bool need_f16 = false;                                 // f as u16 residues: only the vector-ALU form of a round reads it
    for (int r = 0; r < rounds; r++) need_f16 |= !ntru_newton_round_matrix_applies(eng, N, bits_of[r], bits_of[r + 1]);
    for (int64_t o = 0; o < B; o += C) {
      const int64_t n = B - o < C ? B - o : C;
      uint16_t *v = d_fq + o * N;
      if (need_f16) hipLaunchKernelGGL(k_signed_to_u16, elementwise_grid(eng, n * N), dim3(256), 0, eng->stream, d_f + o * N, (long)(n * N),
                                       (u32)q, (u16 *)f16.p);
      for (int r = 0; r < rounds; r++) {
        // Hensel lifting: round r takes v from kb = bits_of[r] to m = bits_of[r + 1] <= 2 kb bits, i.e. runs modulo mr = 2^m (the last
        // one modulo q).  The inverse modulo q is unique.
        const int kb = bits_of[r], m = bits_of[r + 1], mr = 1 << m;
        const long tot = (long)(n * N);
        const long nvec = ((((unsigned long long)v | (unsigned long long)u.p) & 15) == 0) ? tot / 8 : 0;
        if (ntru_newton_round_matrix_applies(eng, N, kb, m)) {
          // Per-item products on the matrix cores, the round in its LIFTED form, ONE kernel.  v is right modulo 2^kb, so
          // f v = 1 + 2^kb e and v (2 - f v) = v - 2^kb (e v) modulo mr = 2^m, m <= 2 kb: only e v modulo 2^(m - kb) is needed, a
          // product of operands of at most kb <= 7 bits (ONE int8 digit plane each: the schedule above guarantees it for log2 q <= 14).
          if (int rc = ntru_launch_newton_round_matrix(eng, N, kb, m, d_f + o * N, v, (long)n))
            return rc == NTRU_NOT_TAKEN ? fail(NTRU_ERR_UNSUPPORTED, "ntru_invert_key_batch: Newton round outside the matrix kernel's range") : rc;
          continue;
        }
        // vector-ALU families: v <- 2 v - f v^2
        if (int rc = ntru_polymul_split_dev(eng, N, mr, v, v, n, (uint16_t *)qs.p, (uint16_t *)t.p)) return rc;
        if (int rc = ntru_polymul_split_dev(eng, N, mr, (const uint16_t *)f16.p, (const uint16_t *)t.p, n, (uint16_t *)qs.p, (uint16_t *)u.p)) return rc;
        if (nvec) hipLaunchKernelGGL(k_newton_combine_vec, elementwise_grid(eng, nvec, true), dim3(256), 0, eng->stream, (u16x8 *)v,
                                     (const u16x8 *)u.p, nvec, (u32)mr);
        if (nvec * 8 < tot) hipLaunchKernelGGL(k_newton_combine, elementwise_grid(eng, tot - nvec * 8), dim3(256), 0, eng->stream, (u16 *)v,
                                               (const u16 *)u.p, nvec * 8, tot, (u32)mr);
      }
      HIP_TRY(hipGetLastError());                          // the next chunk reuses the temporaries in stream order
    }
  }
  if (int rc = join()) return rc;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_invert_key");
  return NTRU_OK;
}

template <bool BIGN, int WAVES, int DR>
static int launch_sampler(ntru_engine *eng, int N, int n1, int n2, int other, const ChaChaKey &ck, uint64_t first_item, int64_t B,
                          uint8_t *d_out) {
  const int NW = (N + 15) / 16;                          // dwords of 16 symbols per row
  const size_t lds = WAVES * ((size_t)64 * NW * 4 + (BIGN ? (size_t)((N + 2) & ~1) * 4 : 0));
  if (lds > 160 * 1024) return fail(NTRU_ERR_UNSUPPORTED, "N too large for the sampler's LDS rows");
  const void *fn = (const void *)k_sample_ternary<BIGN, WAVES, DR>;
  int per_cu = 0;
  if (int rc = ntru_blocks_per_cu(eng, fn, WAVES * 64, lds, &per_cu)) return rc;
  if (eng->max_blocks_per_cu && eng->max_blocks_per_cu < per_cu) per_cu = eng->max_blocks_per_cu;     // NTRU_MAX_BLOCKS_PER_CU (experiments)
  long blocks = (B + WAVES * 64 - 1) / (WAVES * 64), cap = (long)eng->cus * (per_cu < 1 ? 1 : per_cu);
  if (blocks > cap) blocks = cap;
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_sample_ternary");
  hipLaunchKernelGGL((k_sample_ternary<BIGN, WAVES, DR>), dim3((unsigned)blocks), dim3(WAVES * 64), lds, eng->stream, N, n1, n2, (u32)other, ck,
                     (unsigned long long)first_item, (long)B, d_out, NW);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

extern "C" int ntru_sample_ternary_dev(ntru_engine_t *eng, int N, int n1, int n2, int other, const uint32_t *key,
                                       uint64_t first_item, int64_t B, uint8_t *d_out) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0 || N < 1 || n1 < 0 || n2 < 0) return fail(NTRU_ERR_ARG, "negative size");
  if (n1 + n2 > N) return fail(NTRU_ERR_ARG, "The total of 1s and -1s cannot exceed the array length.");   // index.js:463
  if (other < 0 || other > 255) return fail(NTRU_ERR_ARG, "`other` must fit a byte");
  if (!key) return fail(NTRU_ERR_ARG, "ntru_sample_ternary: key is NULL");
  if (B == 0) return NTRU_OK;
  if (!d_out) return fail(NTRU_ERR_ARG, "ntru_sample_ternary: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  ChaChaKey ck;
  memcpy(ck.k, key, 32);
  auto go = [&](auto dr) -> int {
    constexpr int DR = decltype(dr)::value;
    return N + 1 < 2048 ? launch_sampler<false, 4, DR>(eng, N, n1, n2, other, ck, first_item, B, d_out)     // <= 32 KB of rows per wave
                        : launch_sampler<true, 1, DR>(eng, N, n1, n2, other, ck, first_item, B, d_out);
  };
  switch (eng->sampler_rounds) {
    case 8: return go(std::integral_constant<int, 4>{});
    case 12: return go(std::integral_constant<int, 6>{});
    default: return go(std::integral_constant<int, 10>{});
  }
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
  hipLaunchKernelGGL(k_pack<u16>, elementwise_grid(eng, B * os * 4), dim3(256), 0, eng->stream, bits, per, data_len, os, d_data,
                     (long)B, (unsigned long long *)d_out);
  HIP_TRY(hipGetLastError());
  return NTRU_OK;
}

// The same for an array of BYTES (decryptBits' value, quotient2, r, m: values <= 255): packOutput without widening it first.
extern "C" int ntru_pack_bytes_batch_dev(ntru_engine_t *eng, int max_val, int data_len, const uint8_t *d_data, int64_t B,
                                         uint64_t *d_out) {
  if (!eng) return fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0) return fail(NTRU_ERR_ARG, "negative batch size");
  if (max_val > 255) return fail(NTRU_ERR_ARG, "ntru_pack_bytes_batch: max_val must fit a byte");
  int bits, per, al, os;
  if (int rc = ntru_pack_params(max_val, data_len, &bits, &per, &al, &os)) return rc;
  if (B == 0) return NTRU_OK;
  if ((!d_data && data_len) || !d_out) return fail(NTRU_ERR_ARG, "ntru_pack_bytes_batch: NULL buffer");
  HIP_TRY(hipSetDevice(eng->device));
  if (bits == 2) {                                         // ternary rows / decrypted values: 16 values per output dword
    hipLaunchKernelGGL(k_pack_bytes2, elementwise_grid(eng, B * os * 8), dim3(256), 0, eng->stream, data_len, os, d_data, (long)B, (u32 *)d_out);
    HIP_TRY(hipGetLastError());
    return NTRU_OK;
  }
  hipLaunchKernelGGL(k_pack<uint8_t>, elementwise_grid(eng, B * os * 4), dim3(256), 0, eng->stream, bits, per, data_len, os, d_data,
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
