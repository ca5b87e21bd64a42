"""ctypes front-end of the CPU oracle (oracle/ntru_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of ntru_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.

Two layers:
  * thin wrappers over the C functions (variable-length int64 polynomials and the
    fixed-stride uint16/uint8 batch entry points);
  * `OracleNTRU`, which restates the reference's scheme-level orchestration
    (index.js:87-197: which products, which padding, field names and orders of the
    {value, inputs, params} witness objects) on top of those wrappers, so the golden
    JSON captured from the reference can be compared with `==`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXACT, FAITHFUL = 0, 1
ERRORS = {
    1: "Cannot divide by zero polynomial.",      # index.js:360
    2: "No inverse exists for division.",        # index.js:378
    3: "The total of 1s and -1s cannot exceed the array length.",  # index.js:463
    4: "oracle: bad argument",
}


def build(force=False):
    so = os.path.join(_HERE, "libntru_oracle.so")
    src = os.path.join(_HERE, "ntru_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libntru_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_mod_inverse.restype = C.c_int64
        _LIB.orc_mod_inverse.argtypes = [C.c_int64, C.c_int64]
    return _LIB


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64).reshape(-1))


def _p(arr):
    return arr.ctypes.data_as(C.c_void_p)


class OracleError(Exception):
    pass


def _check(rc):
    if rc:
        raise OracleError(ERRORS.get(rc, "oracle error %d" % rc))


# ---- generic polynomial functions (variable length, int64) ---------------------------------

def degree(a):
    a = _i64(a)
    return lib().orc_degree(_p(a), C.c_int(a.size))


def trim(a):
    a = _i64(a).copy()
    if a.size == 0:
        return [0]
    n = lib().orc_trim(_p(a), C.c_int(a.size))
    return a[:n].tolist()


def mod_inverse(a, p):
    v = lib().orc_mod_inverse(int(a), int(p))
    return None if v < 0 else int(v)


def add(a, b, p):
    a, b = _i64(a), _i64(b)
    out = np.zeros(max(a.size, b.size, 1), dtype=np.int64)
    n = lib().orc_add(_p(a), C.c_int(a.size), _p(b), C.c_int(b.size), C.c_int64(p), _p(out))
    return out[:n].tolist()


def multiply(a, b, p, mode=FAITHFUL):
    a, b = _i64(a), _i64(b)
    out = np.zeros(max(a.size + b.size - 1, 1), dtype=np.int64)
    fn = lib().orc_multiply_fft if mode == FAITHFUL else lib().orc_multiply_exact
    n = fn(_p(a), C.c_int(a.size), _p(b), C.c_int(b.size), C.c_int64(p), _p(out))
    return out[:n].tolist()


def divide(a, b, p):
    """Generic long division, index.js:358-401."""
    a, b = _i64(a), _i64(b)
    quot = np.zeros(max(a.size, 1), dtype=np.int64)
    rem = np.zeros(max(a.size, b.size, 1) + 1, dtype=np.int64)
    lq, lr = C.c_int(0), C.c_int(0)
    _check(lib().orc_divide_long(_p(a), C.c_int(a.size), _p(b), C.c_int(b.size), C.c_int64(p),
                                 _p(quot), C.byref(lq), _p(rem), C.byref(lr)))
    return {"quotient": quot[:lq.value].tolist(), "remainder": rem[:lr.value].tolist()}


def divide_by_I(a, N, p):
    """Closed form of divide(a, 1 - x^N, p) for reduced a."""
    a = _i64(a)
    quot = np.zeros(max(N, 1), dtype=np.int64)
    rem = np.zeros(max(a.size, N, 1), dtype=np.int64)
    lq, lr = C.c_int(0), C.c_int(0)
    _check(lib().orc_divide_by_I(_p(a), C.c_int(a.size), C.c_int(N), C.c_int64(p),
                                 _p(quot), C.byref(lq), _p(rem), C.byref(lr)))
    return {"quotient": quot[:lq.value].tolist(), "remainder": rem[:lr.value].tolist()}


def generate_custom_array(length, n1, nm1, draws):
    draws = np.ascontiguousarray(np.asarray(draws, dtype=np.uint32))
    out = np.zeros(max(length, 1), dtype=np.int64)
    _check(lib().orc_generate_custom_array(C.c_int(length), C.c_int(n1), C.c_int(nm1), _p(draws), _p(out)))
    return out[:length].tolist()


def calc_nbits(mod, N):
    return lib().orc_calc_nbits(C.c_int64(mod), C.c_int(N))


def polymul_split(a, b, N, mod, addend=None, mode=EXACT):
    """(a*b mod `mod`) [+ addend], split by 1 - x^N -> trimmed quotient / remainder lists."""
    a, b = _i64(a), _i64(b)
    quot = np.zeros(2 * N + a.size + b.size + 8, dtype=np.int64)
    rem = np.zeros(2 * N + a.size + b.size + 8, dtype=np.int64)
    lq, lr = C.c_int(0), C.c_int(0)
    if addend is None:
        addp, ladd = None, 0
    else:
        add_arr = _i64(addend)
        addp, ladd = _p(add_arr), add_arr.size
        if add_arr.size == 0:  # keep a valid pointer for the empty plaintext
            add_arr = np.zeros(1, dtype=np.int64)
            addp = _p(add_arr)
    _check(lib().orc_polymul_split(_p(a), C.c_int(a.size), _p(b), C.c_int(b.size), addp, C.c_int(ladd),
                                   C.c_int(N), C.c_int64(mod), C.c_int(mode),
                                   _p(quot), C.byref(lq), _p(rem), C.byref(lr)))
    return quot[:lq.value].tolist(), rem[:lr.value].tolist()


def expand(arr, n):
    """index.js:534-536 with fill 0; the reference raises RangeError when arr is longer than n."""
    arr = list(arr)
    if len(arr) > n:
        raise ValueError("Invalid array length")
    return arr + [0] * (n - len(arr))


# ---- fixed-stride batch entry points ---------------------------------------------------------

def _c(a, dt):
    return np.ascontiguousarray(np.asarray(a, dtype=dt))


def polymul_split_batch(N, mod, a, b, mode=EXACT):
    a, b = _c(a, np.uint16).reshape(-1, N), _c(b, np.uint16).reshape(-1, N)
    B = a.shape[0]
    quot, rem = np.zeros((B, N), np.uint16), np.zeros((B, N), np.uint16)
    _check(lib().orc_polymul_split_batch(C.c_int(N), C.c_int(mod), _p(a), _p(b), C.c_int64(B),
                                         _p(quot), _p(rem), C.c_int(mode)))
    return quot, rem


def encrypt_batch(N, q, h, r, m, mode=EXACT, want_quot=True):
    h = _c(h, np.uint16).reshape(N)
    r, m = _c(r, np.uint8).reshape(-1, N), _c(m, np.uint8).reshape(-1, N)
    B = r.shape[0]
    e = np.zeros((B, N), np.uint16)
    quot = np.zeros((B, N), np.uint16) if want_quot else None
    _check(lib().orc_encrypt_batch(C.c_int(N), C.c_int(q), _p(h), _p(r), _p(m), C.c_int64(B), _p(e),
                                   _p(quot) if want_quot else None, C.c_int(mode)))
    return e, quot


def decrypt_batch(N, q, p, f, fp, e, mode=EXACT, want_witness=True):
    f, fp = _c(f, np.int8).reshape(N), _c(fp, np.uint8).reshape(N)
    e = _c(e, np.uint16).reshape(-1, N)
    B = e.shape[0]
    value = np.zeros((B, N), np.uint8)
    if want_witness:
        q1, r1, q2 = np.zeros((B, N), np.uint16), np.zeros((B, N), np.uint16), np.zeros((B, N), np.uint8)
        ptrs = (_p(q1), _p(r1), _p(q2))
    else:
        q1 = r1 = q2 = None
        ptrs = (None, None, None)
    _check(lib().orc_decrypt_batch(C.c_int(N), C.c_int(q), C.c_int(p), _p(f), _p(fp), _p(e), C.c_int64(B),
                                   _p(value), *ptrs, C.c_int(mode)))
    return value, q1, r1, q2


def verify_keys_batch(N, q, p, f, g, fq, fp, h, mode=EXACT):
    f, g = _c(f, np.int8).reshape(-1, N), _c(g, np.int8).reshape(-1, N)
    fq, h = _c(fq, np.uint16).reshape(-1, N), _c(h, np.uint16).reshape(-1, N)
    fp = _c(fp, np.uint8).reshape(-1, N)
    B = f.shape[0]
    out = {
        "quot_fq": np.zeros((B, N), np.uint16), "rem_fq": np.zeros((B, N), np.uint16),
        "quot_fp": np.zeros((B, N), np.uint8), "rem_fp": np.zeros((B, N), np.uint8),
        "quot_h": np.zeros((B, N), np.uint16), "rem_h": np.zeros((B, N), np.uint16),
        "flags": np.zeros(B, np.uint8),
    }
    _check(lib().orc_verify_keys_batch(C.c_int(N), C.c_int(q), C.c_int(p), _p(f), _p(g), _p(fq), _p(fp), _p(h),
                                       C.c_int64(B), _p(out["quot_fq"]), _p(out["rem_fq"]), _p(out["quot_fp"]),
                                       _p(out["rem_fp"]), _p(out["quot_h"]), _p(out["rem_h"]), _p(out["flags"]),
                                       C.c_int(mode)))
    return out


def public_key_batch(N, q, p, fq, g, mode=EXACT):
    """generatePublicKeyH (index.js:72-79): remainder of ((p*fq) % q) * g by I, mod q, untrimmed rows."""
    fq = _c(fq, np.uint16).reshape(-1, N).astype(np.int64)
    g = _c(g, np.int8).reshape(-1, N).astype(np.int64)
    _, rem = polymul_split_batch(N, q, (fq * p) % q, g % q, mode)
    return rem


def chacha20_block(key, counter, nonce, rounds=20):
    """The ChaCha block function: rounds = 20 is RFC 8439's; 12 and 8 are the reduced variants ntru_engine_set_sampler_rounds offers."""
    key = np.ascontiguousarray(np.asarray(key, dtype=np.uint32)); nonce = np.ascontiguousarray(np.asarray(nonce, dtype=np.uint32))
    out = np.zeros(16, np.uint32)
    lib().orc_chacha_block(_p(key), C.c_uint32(counter), _p(nonce), C.c_int(rounds), _p(out))
    return out


def draw_stream(key, item, n):
    key = np.ascontiguousarray(np.asarray(key, dtype=np.uint32))
    out = np.zeros(max(n, 1), np.uint32)
    lib().orc_draw_stream(_p(key), C.c_uint64(item), C.c_int(n), _p(out))
    return out[:n]


def sample_ternary_batch(N, n1, n2, other, key, first_item, B, rounds=20):
    key = np.ascontiguousarray(np.asarray(key, dtype=np.uint32))
    out = np.zeros((B, N), np.uint8)
    _check(lib().orc_sample_ternary_batch_rounds(C.c_int(N), C.c_int(n1), C.c_int(n2), C.c_int(other), _p(key),
                                                 C.c_uint64(first_item), C.c_int64(B), C.c_int(rounds), _p(out)))
    return out


def pack_params(max_val, data_len):
    v = [C.c_int(0) for _ in range(4)]
    _check(lib().orc_pack_params(C.c_int(max_val), C.c_int(data_len), *[C.byref(x) for x in v]))
    return dict(zip(("maxInputBits", "numInputsPerOutput", "arrLen", "outputSize"), (x.value for x in v)))


def limbs_to_ints(limbs):
    """[..., 4] little-endian uint64 limbs -> Python ints."""
    limbs = np.asarray(limbs, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(w) << (64 * k) for k, w in enumerate(row)) for row in limbs]


def ints_to_limbs(vals):
    out = np.zeros((len(vals), 4), np.uint64)
    for i, v in enumerate(vals):
        for k in range(4):
            out[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def pack_batch(max_val, data_len, data):
    data = _c(data, np.uint16).reshape(-1, data_len) if data_len else np.zeros((len(data), 0), np.uint16)
    B = data.shape[0]
    pr = pack_params(max_val, data_len)
    out = np.zeros((B, pr["outputSize"], 4), np.uint64)
    _check(lib().orc_pack_batch(C.c_int(max_val), C.c_int(data_len), _p(data), C.c_int64(B), _p(out)))
    return out


def unpack_batch(max_val, packed_bits, limbs):
    limbs = np.ascontiguousarray(np.asarray(limbs, dtype=np.uint64))
    B, S = limbs.shape[0], limbs.shape[1]
    bits = pack_params(max_val, 0)["maxInputBits"]
    per = packed_bits // bits
    out = np.zeros((B, S * per), np.uint16)
    _check(lib().orc_unpack_batch(C.c_int(max_val), C.c_int(packed_bits), _p(limbs), C.c_int(S), C.c_int64(B), _p(out)))
    return out


# ---- scheme-level restatement ----------------------------------------------------------------

class OracleNTRU:
    """index.js:7-207, hot-path methods only (no key generation).  Plain Python lists in and out."""

    def __init__(self, N=167, p=3, q=128, df=61, dg=20, dr=18, f=None, fp=None, fq=None, g=None, h=None,
                 mode=EXACT):
        self.N, self.p, self.q, self.df, self.dg, self.dr = N, p, q, df, dg, dr
        self.f, self.fp, self.fq, self.g, self.h = f, fp, fq, g, h
        self.mode = mode
        self.I = [1] + [0] * (N - 1) + [-1]           # index.js:25-27

    def calculate_nq(self):
        return calc_nbits(self.q, self.N)

    def calculate_np(self):
        return calc_nbits(self.p, self.N)

    def encrypt_bits(self, m, r_signed):
        """index.js:87-110 with the sampler's output r_signed in {-1,0,1} supplied by the caller."""
        N, q = self.N, self.q
        r = [self.p - 1 if x == -1 else x for x in r_signed]          # :89
        quot, rem = polymul_split(r, self.h, N, q, addend=m, mode=self.mode)   # :90-92
        return {
            "value": trim(rem),
            "inputs": {
                "r": r,
                "m": expand(m, N),
                "h": expand(self.h, N),
                "quotientE": expand([x % q for x in quot], N + 1),
                "remainderE": expand(rem, N + 1),
            },
            "params": [q, self.calculate_nq(), N],
        }

    def decrypt_bits(self, e):
        """index.js:111-140."""
        N, q, p = self.N, self.q, self.p
        f = [q - 1 if x == -1 else x for x in self.f]                   # :112
        quot1, rem1 = polymul_split(f, e, N, q, mode=self.mode)         # :113-114
        lifted = [(x + 1) % p if x > q / 2 else x % p for x in rem1]    # :117
        quot2, rem2 = polymul_split(self.fp, lifted, N, p, mode=self.mode)  # :118-119
        return {
            "value": trim(rem2),
            "inputs": {
                "f": expand(f, N),
                "fp": expand(self.fp, N),
                "e": expand(e, N),
                "quotient1": expand(quot1, N + 1),
                "remainder1": expand(rem1, N + 1),
                "quotient2": expand(quot2, N + 1),
                "remainder2": expand(rem2, N + 1),
            },
            "params": [q, self.calculate_nq(), p, self.calculate_np(), N],
        }

    def verify_keys_inputs(self):
        """index.js:141-197."""
        for attr, msg in (("f", "missing private key F"), ("fq", "missing private key Fq"),
                          ("fp", "missing private key Fp"), ("g", "missing private key G"),
                          ("h", "missing public key H")):
            if not getattr(self, attr):
                raise OracleError(msg)
        N, q, p = self.N, self.q, self.p
        nq, np_ = self.calculate_nq(), self.calculate_np()
        fmodq = [q - 1 if x == -1 else x for x in self.f]
        fmodp = [p - 1 if x == -1 else x for x in self.f]
        fqp = [x * p for x in self.fq]                                   # :155, unreduced
        g = [q - 1 if x == -1 else x for x in self.g]
        qfq, rfq = polymul_split(self.fq, fmodq, N, q, mode=self.mode)
        if len(rfq) != 1 and rfq[0] != 1:
            raise OracleError("invalid fq")
        qfp, rfp = polymul_split(self.fp, fmodp, N, p, mode=self.mode)
        if len(rfp) != 1 and rfp[0] != 1:
            raise OracleError("invalid fp")
        qh, rh = polymul_split(fqp, g, N, q, mode=self.mode)
        if any(i >= len(rh) or rh[i] != cur for i, cur in enumerate(self.h)):
            raise OracleError("invalid h")

        def case(params, f_, fq_, quot, rem):
            return {"params": params, "inputs": {"f": expand(f_, N), "fq": expand(fq_, N),
                                                 "quotientI": expand(quot, N + 1),
                                                 "remainderI": expand(rem, N + 1)}}
        return {
            "fq": case([q, nq, N], fmodq, self.fq, qfq, rfq),
            "fp": case([p, np_, N], fmodp, self.fp, qfp, rfp),
            "h": case([q, nq, N], g, fqp, qh, rh),
        }
