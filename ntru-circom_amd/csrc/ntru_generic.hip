// ntru_generic.hip -- the reference-faithful GENERIC family: polynomial arithmetic exactly as numtel/ntru-circom's
// index.js performs it, for the calls the fast kernels of ntru_engine.hip do not cover:
//   multiplyPolynomials (index.js:319-355)          any modulus 1 .. 2^26 (e.g. the 2^20 of test/circuits.test.js:72)
//   dividePolynomials   (index.js:358-401)          ANY divisor: long division, per-step inverse of the leading coefficient
//   extendedEuclideanAlgorithm (index.js:425-459)   incl. its `&&` gcd check and the un-normalised intermediate values
//   polyInv             (index.js:491-514)          EEA modulo 2 + the `exponent - 1` Newton rounds, or plain EEA
// Coefficients are int64 and follow JS Number semantics where those are exact (|values| and modulus <= 2^26, so every
// product stays below 2^53): `%` is the truncated remainder (sign of the dividend), inputs may be negative or unreduced,
// array LENGTHS are tracked because the reference's control flow depends on them (`r0.length !== 1 && r0[0] !== 1`).
// The one thing a JS array can hold that an int64 cannot is -0 (e.g. `-2 % 2`); it is returned as 0.
//
// One item per wavefront (block = 64 threads).  Polynomials live in a per-item global work area; the lanes cooperate on
// the O(length) inner loops (degree scans with ballots, the subtraction of coeff * divisor, one output coefficient of a
// product per lane); control flow is wave-uniform.  This is a latency-oriented kernel for single calls and small
// batches -- key generation at throughput runs on k_invert_key / k_polymul_m instead (ntru_invert_key_batch).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>

#include "engine_internal.h"

typedef long long i64;

namespace {

constexpr i64 GEN_MAX = (i64)1 << 26;      // bound on modulus and |coefficients| (products exact in a JS double)

enum : int { ST_OK = 0, ST_DIV_ZERO = NTRU_GENERIC_DIV_BY_ZERO, ST_NO_INVERSE = NTRU_GENERIC_NO_INVERSE,
             ST_INVALID_GCD = NTRU_GENERIC_INVALID_GCD, ST_CAPACITY = NTRU_GENERIC_CAPACITY };

struct Poly { i64 *c; int len; };          // wave-uniform descriptor of an array in the work area

__device__ __forceinline__ int lane() { return (int)threadIdx.x; }

// Orders this wave's global writes before its later reads of them (block = one wave; the barrier is a formality).
__device__ __forceinline__ void phase() { __syncthreads(); }

// index.js:210-215: highest index with a non-zero coefficient among the first `len`, -1 if none.
__device__ int g_degree(const i64 *c, int len) {
  for (int base = ((len - 1) >> 6) << 6; base >= 0 && len > 0; base -= 64) {
    const int idx = base + lane();
    const unsigned long long m = __ballot(idx < len && c[idx] != 0);
    if (m) return base + 63 - __clzll(m);
  }
  return -1;
}

// index.js:218-221 applied in place: the length becomes degree + 1, or the polynomial becomes [0].
__device__ void g_trim(Poly &a) {
  const int d = g_degree(a.c, a.len);
  if (d >= 0) { a.len = d + 1; return; }
  if (lane() == 0) a.c[0] = 0;
  a.len = 1;
  phase();
}

// index.js:224-232: the x in [1, p) with (a * x) % p == 1 after a = ((a % p) + p) % p, or 0 for `null`.  The reference
// searches x upwards; the inverse is unique when it exists, so the extended Euclidean algorithm on integers gives the
// same value.
__device__ i64 g_mod_inverse(i64 a, i64 p) {
  a = ((a % p) + p) % p;
  if (p == 1 || a == 0) return 0;
  i64 r0 = p, r1 = a, t0 = 0, t1 = 1;
  while (r1 != 0) {
    const i64 qq = r0 / r1;
    i64 t = r0 - qq * r1; r0 = r1; r1 = t;
    t = t0 - qq * t1; t0 = t1; t1 = t;
  }
  if (r0 != 1) return 0;
  return ((t0 % p) + p) % p;
}

__device__ void g_copy(Poly &dst, const Poly &src) {
  for (int i = lane(); i < src.len; i += 64) dst.c[i] = src.c[i];
  dst.len = src.len;
  phase();
}

// index.js:319-355 with the FFT replaced by the exact sum it rounds to: every coefficient into [0, p), trimmed.
__device__ int g_multiply(Poly &out, const Poly &a, const Poly &b, i64 p, int cap) {
  if (a.len == 0 || b.len == 0) {
    if (lane() == 0) out.c[0] = 0;
    out.len = 1;
    phase();
    return ST_OK;
  }
  const int rl = a.len + b.len - 1;
  if (rl > cap) return ST_CAPACITY;
  for (int k = lane(); k < rl; k += 64) {
    const int lo = k - (b.len - 1) > 0 ? k - (b.len - 1) : 0, hi = k < a.len - 1 ? k : a.len - 1;
    // |term| < 2^52 (check_generic: |coefficients| <= 2^26), so 256 terms stay below 2^60; every run of 256 is reduced before it
    // joins the total (|run % p| < 2^26, <= 16 runs): the int64 sum can no longer wrap at lengths up to 4096, and the residue of
    // the exact sum -- what the reference's rounded FFT product reduces -- is unchanged.
    i64 s = 0;
    for (int i0 = lo; i0 <= hi; i0 += 256) {
      const int i1 = i0 + 255 < hi ? i0 + 255 : hi;
      i64 t = 0;
      for (int i = i0; i <= i1; i++) t += a.c[i] * b.c[k - i];
      s += t % p;
    }
    out.c[k] = ((s % p) + p) % p;
  }
  out.len = rl;
  phase();
  g_trim(out);
  return ST_OK;
}

// index.js:247-256
__device__ int g_subtract(Poly &out, const Poly &a, const Poly &b, i64 p, int cap) {
  const int n = a.len > b.len ? a.len : b.len;
  if (n > cap) return ST_CAPACITY;
  for (int i = lane(); i < n; i += 64) {
    const i64 x = i < a.len ? a.c[i] : 0, y = i < b.len ? b.c[i] : 0;
    out.c[i] = (((x - y) % p) + p) % p;
  }
  out.len = n;
  phase();
  g_trim(out);
  return ST_OK;
}

// index.js:404-406: poly.map(c => (c * scalar) % p), no normalisation, no trimming.  In place when out.c == a.c.
__device__ void g_scale(Poly &out, const Poly &a, i64 s, i64 p) {
  for (int i = lane(); i < a.len; i += 64) out.c[i] = (a.c[i] * s) % p;
  out.len = a.len;
  phase();
}

// index.js:358-401.  quot and rem (the working copy of the dividend) are fresh arrays of capacity `cap`.
__device__ int g_divide(Poly &quot, Poly &rem, const Poly &a, const Poly &b, i64 p, int cap) {
  const int db = g_degree(b.c, b.len);
  if (db == -1) return ST_DIV_ZERO;
  if (a.len > cap) return ST_CAPACITY;
  g_copy(rem, a);
  int dg = g_degree(rem.c, rem.len);
  const int qlen = dg - db + 1 > 0 ? dg - db + 1 : 0;
  for (int i = lane(); i < (qlen > 1 ? qlen : 1); i += 64) quot.c[i] = 0;
  quot.len = qlen;
  phase();
  const i64 lead_b = b.c[db];
  i64 inv = 0;
  bool have_inv = false;
  while (dg >= db) {
    if (!have_inv) {                                   // the reference recomputes it every step: same value each time
      inv = g_mod_inverse(lead_b, p);
      if (inv == 0) return ST_NO_INVERSE;
      have_inv = true;
    }
    const i64 coeff = (rem.c[dg] * inv) % p;           // JS %: may be negative for a negative dividend coefficient
    const int diff = dg - db;
    if (lane() == 0) quot.c[diff] = coeff;
    for (int i = lane(); i <= db; i += 64) {
      i64 v = (rem.c[i + diff] - coeff * b.c[i]) % p;
      if (v < 0) v += p;
      rem.c[i + diff] = v;
    }
    phase();
    const int next = g_degree(rem.c, dg + 1);
    if (next >= dg) return ST_CAPACITY;                 // cannot happen (the leading term cancels); never spin on the GPU
    dg = next;
  }
  g_trim(quot);
  g_trim(rem);
  return ST_OK;
}

// Work area of one item: NBUF arrays of `cap` coefficients.
constexpr int NBUF = 10;

struct Work {
  i64 *base; int cap; unsigned used;
  __device__ i64 *take() {                              // wave-uniform; ten buffers always suffice (see g_eea / g_poly_inv)
    for (int i = 0; i < NBUF; i++) if (!(used >> i & 1u)) { used |= 1u << i; return base + (size_t)i * cap; }
    return base;
  }
  __device__ void give(const i64 *p) { used &= ~(1u << (unsigned)((p - base) / cap)); }
};

// index.js:425-459.  gcd / inverse come back in buffers of `w`.
__device__ int g_eea(Poly &gcd, Poly &inverse, const Poly &a, const Poly &b, i64 p, Work &w) {
  const int cap = w.cap;
  if (a.len > cap || b.len > cap) return ST_CAPACITY;
  Poly r0{w.take(), 0}, r1{w.take(), 0}, s0{w.take(), 1}, s1{w.take(), 1};
  g_copy(r0, a);
  g_copy(r1, b);
  if (lane() == 0) { s0.c[0] = 1; s1.c[0] = 0; }
  phase();
  while (g_degree(r1.c, r1.len) >= 0) {
    Poly q{w.take(), 0}, rem{w.take(), 0};
    if (int st = g_divide(q, rem, r0, r1, p, cap)) return st;
    w.give(r0.c);
    r0 = r1; r1 = rem;
    Poly t{w.take(), 0}, u{w.take(), 0};
    if (int st = g_multiply(t, q, s1, p, cap)) return st;
    if (int st = g_subtract(u, s0, t, p, cap)) return st;
    w.give(q.c); w.give(t.c); w.give(s0.c);
    s0 = s1; s1 = u;
  }
  const int d0 = g_degree(r0.c, r0.len);
  const i64 inv = d0 >= 0 ? g_mod_inverse(r0.c[d0], p) : 0;      // modInverse(undefined) is null as well
  if (inv != 0 && inv != 1) { g_scale(r0, r0, inv, p); g_scale(s0, s0, inv, p); }
  const bool first_is_one = r0.len > 0 && r0.c[0] == 1;
  if (r0.len != 1 && !first_is_one) return ST_INVALID_GCD;        // the reference's `&&` (index.js:451)
  w.give(r1.c); w.give(s1.c);
  gcd = r0; inverse = s0;
  return ST_OK;
}

// index.js:491-514
__device__ int g_poly_inv(Poly &inverse, const Poly &a, const Poly &I, i64 mod, Work &w) {
  const int cap = w.cap;
  if ((mod & (mod - 1)) != 0) {                        // not a power of two: plain EEA (index.js:509-512)
    Poly g{nullptr, 0};
    return g_eea(g, inverse, a, I, mod, w);
  }
  Poly g{nullptr, 0}, inv{nullptr, 0};
  if (int st = g_eea(g, inv, a, I, 2, w)) return st;
  w.give(g.c);
  int e = 0;
  while (((i64)1 << e) < mod) e++;
  for (int k = 1; k < e; k++) {
    Poly twice{w.take(), 0}, sq{w.take(), 0}, cube{w.take(), 0}, upd{w.take(), 0}, q{w.take(), 0}, rem{w.take(), 0};
    g_scale(twice, inv, 2, mod);
    if (int st = g_multiply(sq, inv, inv, mod, cap)) return st;
    if (int st = g_multiply(cube, a, sq, mod, cap)) return st;
    if (int st = g_subtract(upd, twice, cube, mod, cap)) return st;
    if (int st = g_divide(q, rem, upd, I, mod, cap)) return st;
    w.give(twice.c); w.give(sq.c); w.give(cube.c); w.give(upd.c); w.give(q.c); w.give(inv.c);
    inv = rem;                                          // already trimmed (index.js:505)
  }
  inverse = inv;
  return ST_OK;
}

__device__ void g_store(i64 *row, int *len_out, const Poly &p, int item) {
  for (int i = lane(); i < p.len; i += 64) row[i] = p.c[i];
  if (lane() == 0) len_out[item] = p.len;
}

// op: 0 multiply, 1 divide, 2 extendedEuclideanAlgorithm, 3 polyInv.  out0 / out1 rows have a pitch of `cap` elements.
__global__ __launch_bounds__(64) void k_generic(int op, int la, int lb, i64 mod, const i64 *__restrict__ a,
                                                const i64 *__restrict__ b, long B, i64 *work, int cap, i64 *out0,
                                                int *len0, i64 *out1, int *len1, unsigned char *status) {
  for (long item = blockIdx.x; item < B; item += gridDim.x) {
    Work w{work + (size_t)blockIdx.x * NBUF * cap, cap, 0u};
    Poly pa{w.take(), 0}, pb{w.take(), 0};
    Poly src_a{const_cast<i64 *>(a) + (size_t)item * la, la}, src_b{const_cast<i64 *>(b) + (size_t)item * lb, lb};
    g_copy(pa, src_a);
    g_copy(pb, src_b);
    int st = ST_OK;
    Poly r0{nullptr, 0}, r1{nullptr, 0};
    if (op == 0) {
      r0 = Poly{w.take(), 0};
      st = g_multiply(r0, pa, pb, mod, cap);
    } else if (op == 1) {
      r0 = Poly{w.take(), 0}; r1 = Poly{w.take(), 0};
      st = g_divide(r0, r1, pa, pb, mod, cap);
    } else if (op == 2) {
      st = g_eea(r0, r1, pa, pb, mod, w);
    } else {
      st = g_poly_inv(r0, pa, pb, mod, w);
    }
    if (st == ST_OK) {
      g_store(out0 + (size_t)item * cap, len0, r0, (int)item);
      if (out1 && r1.c) g_store(out1 + (size_t)item * cap, len1, r1, (int)item);
    } else if (lane() == 0) {
      len0[item] = 0;
      if (len1) len1[item] = 0;
    }
    if (lane() == 0) status[item] = (unsigned char)st;
    phase();
  }
}

int check_generic(ntru_engine *eng, int la, int lb, int64_t mod, int64_t B) {
  if (!eng) return ntru_fail(NTRU_ERR_ARG, "engine is NULL");
  if (B < 0 || la < 0 || lb < 0) return ntru_fail(NTRU_ERR_ARG, "negative size");
  if (la > 4096 || lb > 4096) return ntru_fail(NTRU_ERR_UNSUPPORTED, "generic family: at most 4096 coefficients per operand");
  if (mod < 1 || mod > GEN_MAX) return ntru_fail(NTRU_ERR_UNSUPPORTED, "generic family: modulus must be in 1 .. 2^26");
  return NTRU_OK;
}

// Host-pointer driver: inputs are range-checked (the exactness bound), staged, one launch, results copied back.
int run_generic(ntru_engine *eng, int op, int la, int lb, int64_t mod, const int64_t *a, const int64_t *b, int64_t B,
                int64_t *out0, int32_t *len0, int64_t *out1, int32_t *len1, uint8_t *status) {
  if (int rc = check_generic(eng, la, lb, mod, B)) return rc;
  if (B == 0) return NTRU_OK;
  if ((la && !a) || (lb && !b) || !out0 || !len0 || !status) return ntru_fail(NTRU_ERR_ARG, "generic family: NULL buffer");
  for (int64_t i = 0; i < B * la; i++)
    if (a[i] > GEN_MAX || a[i] < -GEN_MAX) return ntru_fail(NTRU_ERR_UNSUPPORTED, "generic family: |coefficient| above 2^26");
  for (int64_t i = 0; i < B * lb; i++)
    if (b[i] > GEN_MAX || b[i] < -GEN_MAX) return ntru_fail(NTRU_ERR_UNSUPPORTED, "generic family: |coefficient| above 2^26");
  HIP_TRY(hipSetDevice(eng->device));
  const int cap = ntru_generic_capacity(la, lb);
  long blocks = (long)eng->cus * 2;
  if (blocks > B) blocks = B;
  const size_t in_a = (size_t)B * la * 8, in_b = (size_t)B * lb * 8, rows = (size_t)B * cap * 8, lens = (size_t)B * 4;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_a = 0, o_b = o_a + up(in_a), o_w = o_b + up(in_b), o_0 = o_w + up((size_t)blocks * NBUF * cap * 8),
               o_1 = o_0 + up(rows), o_l0 = o_1 + up(rows), o_l1 = o_l0 + up(lens), o_st = o_l1 + up(lens),
               total = o_st + up((size_t)B);
  // the engine's shared scratch buffer, ordered against its last user on another stream (key inversion on stream A, then this
  // call after ntru_engine_set_stream(B): B waits for A's event before the first copy below lands in the buffer)
  ScratchHold hold(eng, total);
  if (hold.rc) return hold.rc;
  char *const d = hold.p;
  hipStream_t s = eng->stream;
  if (in_a) HIP_TRY(hipMemcpyAsync(d + o_a, a, in_a, hipMemcpyHostToDevice, s));
  if (in_b) HIP_TRY(hipMemcpyAsync(d + o_b, b, in_b, hipMemcpyHostToDevice, s));
  snprintf(eng->last_kernel, sizeof eng->last_kernel, "k_generic");
  hipLaunchKernelGGL(k_generic, dim3((unsigned)blocks), dim3(64), 0, s, op, la, lb, (i64)mod, (const i64 *)(d + o_a),
                     (const i64 *)(d + o_b), (long)B, (i64 *)(d + o_w), cap, (i64 *)(d + o_0), (int *)(d + o_l0),
                     out1 ? (i64 *)(d + o_1) : nullptr, out1 ? (int *)(d + o_l1) : nullptr, (unsigned char *)(d + o_st));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out0, d + o_0, rows, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(len0, d + o_l0, lens, hipMemcpyDeviceToHost, s));
  if (out1) {
    HIP_TRY(hipMemcpyAsync(out1, d + o_1, rows, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(len1, d + o_l1, lens, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipMemcpyAsync(status, d + o_st, (size_t)B, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return NTRU_OK;
}

}  // namespace

extern "C" int ntru_generic_capacity(int la, int lb) { return la + 2 * lb + 8; }

extern "C" int ntru_generic_multiply(ntru_engine_t *eng, int la, int lb, int64_t mod, const int64_t *a, const int64_t *b,
                                     int64_t B, int64_t *out, int32_t *out_len) {
  if (!eng) return ntru_fail(NTRU_ERR_ARG, "engine is NULL");
  if (B > 0 && !out_len) return ntru_fail(NTRU_ERR_ARG, "generic family: NULL buffer");
  // no status can be raised by a product; a scratch byte per item keeps the kernel interface uniform
  std::string st((size_t)(B > 0 ? B : 0), '\0');
  return run_generic(eng, 0, la, lb, mod, a, b, B, out, out_len, nullptr, nullptr, (uint8_t *)&st[0]);
}

extern "C" int ntru_generic_divide(ntru_engine_t *eng, int la, int lb, int64_t mod, const int64_t *a, const int64_t *b,
                                   int64_t B, int64_t *quot, int32_t *quot_len, int64_t *rem, int32_t *rem_len,
                                   uint8_t *status) {
  if (B > 0 && (!rem || !rem_len)) return ntru_fail(NTRU_ERR_ARG, "generic family: NULL buffer");
  return run_generic(eng, 1, la, lb, mod, a, b, B, quot, quot_len, rem, rem_len, status);
}

extern "C" int ntru_generic_eea(ntru_engine_t *eng, int la, int lb, int64_t mod, const int64_t *a, const int64_t *b,
                                int64_t B, int64_t *gcd, int32_t *gcd_len, int64_t *inverse, int32_t *inverse_len,
                                uint8_t *status) {
  if (B > 0 && (!inverse || !inverse_len)) return ntru_fail(NTRU_ERR_ARG, "generic family: NULL buffer");
  return run_generic(eng, 2, la, lb, mod, a, b, B, gcd, gcd_len, inverse, inverse_len, status);
}

extern "C" int ntru_generic_poly_inv(ntru_engine_t *eng, int la, int lb, int64_t mod, const int64_t *a,
                                     const int64_t *poly_i, int64_t B, int64_t *inverse, int32_t *inverse_len,
                                     uint8_t *status) {
  return run_generic(eng, 3, la, lb, mod, a, poly_i, B, inverse, inverse_len, nullptr, nullptr, status);
}
