"""Host-side mirror of the reference's hot-path interface (numtel/ntru-circom index.js) over the HIP engine.

Same names, argument meaning, return shapes and error behaviour as the reference for:
  class NTRU                 index.js:7-207   (constructor options, public fields, encryptBits, decryptBits,
                                               verifyKeysInputs, encryptStr, decryptStr, calculateNq/Np)
  multiplyPolynomials        index.js:319-355
  addPolynomials             index.js:235-244 (O(N) host glue, as in the reference)
  degree / trimPolynomial / expandArray / generateCustomArray / stringToBits / bitsToString

All polynomial products and the quotient/remainder split run on the GPU through the C ABI
(include/ntru_engine.h); this file only reshapes data (padding, trimming, dict building), which is what
SURVEY.md section 8 assigns to the shim.  Key generation (index.js:30-79) runs on the device as well; polyInv,
extendedEuclideanAlgorithm, dividePolynomials by an arbitrary divisor and products modulo more than 65536 run on the
engine's generic family (ntru_generic_*), which follows the reference step by step, non-units included.
"""
import secrets

import numpy as np

from .engine import (FLAG_INVALID_FP, FLAG_INVALID_FQ, FLAG_INVALID_H, FLAG_NOT_UNIT_MOD2, FLAG_NOT_UNIT_MODP,
                     GENERIC_ERRORS, Engine, EngineError)

_DEFAULT_ENGINE = None


def default_engine():
    """Engine on HIP device 0, created on first use.  Raises EngineError when no GPU / library is present."""
    global _DEFAULT_ENGINE
    if _DEFAULT_ENGINE is None:
        _DEFAULT_ENGINE = Engine(0)
    return _DEFAULT_ENGINE


# ---- small helpers with the reference's semantics --------------------------------------------------------

def degree(poly):
    """index.js:210-215"""
    for i in range(len(poly) - 1, -1, -1):
        if poly[i] != 0:
            return i
    return -1


def trimPolynomial(poly):
    """index.js:218-221"""
    d = degree(poly)
    return list(poly[:d + 1]) if d >= 0 else [0]


def expandArray(arr, length, fill=0):
    """index.js:534-536; the reference's `Array(len - arr.length)` throws RangeError when arr is too long."""
    if len(arr) > length:
        raise ValueError("Invalid array length")
    return list(arr) + [fill] * (length - len(arr))


def addPolynomials(a, b, p):
    """index.js:235-244"""
    n = max(len(a), len(b))
    out = [((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % p for i in range(n)]
    return trimPolynomial(out)


def generateCustomArray(length, numOnes, numNegOnes, rand_u32=None):
    """index.js:461-488: numOnes 1s, numNegOnes -1s, rest 0, Fisher-Yates with i descending and
    j = u32 % (i+1), exactly length-1 draws.  `rand_u32` (callable -> uint32) defaults to the OS CSPRNG."""
    if numOnes + numNegOnes > length:
        raise ValueError("The total of 1s and -1s cannot exceed the array length.")
    rand_u32 = rand_u32 or (lambda: secrets.randbits(32))
    arr = [1] * numOnes + [-1] * numNegOnes + [0] * (length - numOnes - numNegOnes)
    for i in range(length - 1, 0, -1):
        j = rand_u32() % (i + 1)
        arr[i], arr[j] = arr[j], arr[i]
    return arr


def stringToBits(s):
    """index.js:538-546"""
    return [int(c) for ch in s for c in format(ord(ch), "08b")]


def bitsToString(bits):
    """index.js:548-556.  JS parseInt(str, 2) reads the longest leading run of binary digits and gives NaN when
    there is none; String.fromCharCode(NaN) is '\\u0000' -- wrong-key decryptions (ternary garbage) rely on that."""
    out = []
    for i in range(0, len(bits), 8):
        digits = "".join(str(b) for b in bits[i:i + 8])
        k = 0
        while k < len(digits) and digits[k] in "01":
            k += 1
        out.append(chr(int(digits[:k], 2) & 0xFFFF if k else 0))
    return "".join(out)


class ReferenceError_(ValueError):
    """An error the reference itself throws (`throw new Error(msg)`): str(e) is the reference's message."""


def _js_mod(x, p):
    """JS `%`: truncated remainder, sign of the dividend."""
    return int(np.fmod(x, p))


def _fast_modulus(N, p):
    return p >= 2 and (p <= 65536 if p & (p - 1) == 0 else N * (p - 1) * (p - 1) < 65536)


def _is_I(b, p):
    N = len(b) - 1
    return N >= 1 and b[0] % p == 1 and b[N] % p == p - 1 and all(x % p == 0 for x in b[1:N])


def _raise_status(st):
    if st:
        raise ReferenceError_(GENERIC_ERRORS[int(st)])


def modInverse(a, p):
    """index.js:224-232 (brute force, like the reference; None when there is no inverse)"""
    a = ((_js_mod(a, p)) + p) % p
    for x in range(1, p):
        if (a * x) % p == 1:
            return x
    return None


def subtractPolynomials(a, b, p):
    """index.js:247-256"""
    n = max(len(a), len(b))
    return trimPolynomial([((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % p for i in range(n)])


def multiplyPolynomialsByScalar(poly, scalar, p):
    """index.js:404-406: no normalisation (JS `%` keeps the sign), no trimming"""
    return [_js_mod(c * scalar, p) for c in poly]


def bigintToBits(value):
    """index.js:558-566: least significant bit first, [] for 0"""
    bits = []
    while value > 0:
        bits.append(value & 1)
        value >>= 1
    return bits


def bitsToBigInt(bits):
    """index.js:568-570: BigInt('0b' + bits.join(''))"""
    return int("".join(str(b) for b in bits), 2)


def multiplyPolynomials(a, b, p, engine=None):
    """index.js:319-355: linear product, each coefficient in [0,p), trailing zeros trimmed.

    Runs on the GPU as one polymul-split in a ring large enough to hold both operands; the linear
    product is recovered from (quotient, remainder): c[N+k] = -quot[k], c[k] = rem[k] - c[N+k].  Moduli the packed
    kernels do not take (above 65536: test/circuits.test.js:72 uses 2^20) run on the generic family."""
    if len(a) == 0 or len(b) == 0:
        return [0]
    eng = engine or default_engine()
    N = max(len(a), len(b), 2)
    if not _fast_modulus(N, p):
        return eng.generic_multiply(list(a), list(b), p)[0]
    ar = np.array([x % p for x in expandArray(a, N)], dtype=np.int64)
    br = np.array([x % p for x in expandArray(b, N)], dtype=np.int64)
    quot, rem = eng.polymul_split(N, p, ar.astype(np.uint16), br.astype(np.uint16))
    hi = (p - quot[0].astype(np.int64)) % p
    lo = (rem[0].astype(np.int64) - hi) % p
    return trimPolynomial(lo.tolist() + hi.tolist()[:N - 1])


def dividePolynomials(a, b, p, engine=None):
    """index.js:358-401.  The hot path's divisor b = I = [1, 0, ..., 0, -1] with a reduced dividend runs the elementwise
    split kernel (ntru_split_by_I; inside encrypt/decrypt/verify the split is fused into the product kernel); any other
    divisor is long division on the generic family, errors included."""
    if degree(b) == -1:
        raise ReferenceError_("Cannot divide by zero polynomial.")
    eng = engine or default_engine()
    N = len(b) - 1
    if not (2 <= p <= 65536 and _is_I(b, p) and len(a) <= 2 * N and all(0 <= x < p for x in a)):
        quot, rem, st = eng.generic_divide(list(a), list(b), p)
        _raise_status(st[0])
        return {"quotient": quot[0], "remainder": rem[0]}
    quot, rem = eng.split_by_I(N, p, [expandArray(a, 2 * N)])
    nq = max(len(a) - N, 0)
    quotient = trimPolynomial(quot[0].tolist()[:nq]) if nq else [0]
    return {"quotient": quotient, "remainder": trimPolynomial(rem[0].tolist())}


def extendedEuclideanAlgorithm(a, b, p, engine=None):
    """index.js:425-459 -> {'gcd', 'inverse'}"""
    eng = engine or default_engine()
    gcd, inv, st = eng.generic_eea(list(a), list(b), p)
    _raise_status(st[0])
    return {"gcd": gcd[0], "inverse": inv[0]}


def polyInv(polyIn, polyI, polyMod, engine=None):
    """index.js:491-514.  Ternary f, polyI = 1 - x^N and modulus 3 or a power of two take the batched inversion kernels
    (the inverse of a unit is unique); everything else, and every f those kernels flag, runs the reference's own sequence
    on the generic family."""
    eng = engine or default_engine()
    N = len(polyI) - 1
    pow2 = 2 <= polyMod <= 65536 and polyMod & (polyMod - 1) == 0
    if (len(polyIn) <= N and all(x in (-1, 0, 1) for x in polyIn) and 2 <= N and eng.supports(N, 2) and
            (polyMod == 3 or pow2) and polyI[0] == 1 and polyI[N] == -1 and all(x == 0 for x in polyI[1:N])):
        f = [expandArray(list(polyIn), N)]
        if polyMod == 3:
            _, fp, flags = eng.invert_key_batch(N, 2, 3, f, want_fq=False)
            if not int(flags[0]) & FLAG_NOT_UNIT_MODP:
                return trimPolynomial(fp[0].tolist())
        else:
            fq, _, flags = eng.invert_key_batch(N, polyMod, 3, f, want_fp=False)
            if not int(flags[0]) & FLAG_NOT_UNIT_MOD2:
                return trimPolynomial(fq[0].tolist())
    inv, st = eng.generic_poly_inv(list(polyIn), list(polyI), polyMod)
    _raise_status(st[0])
    return inv[0]


def addCiphertexts(e1, e2, q, engine=None):
    """addPolynomials on the GPU (index.js:235-244) for operands already in [0,q): the homomorphic sum of
    test/reference.test.js:46-61."""
    eng = engine or default_engine()
    N = max(len(e1), len(e2), 1)
    out = eng.add_batch(N, q, [expandArray(e1, N)], [expandArray(e2, N)])
    return trimPolynomial(out[0].tolist())


def packOutput(maxVal, dataLen, data, engine=None):
    """index.js:572-596: same dict as the reference; `expected` is a list of Python ints (the reference's BigInts)."""
    eng = engine or default_engine()
    pr = eng.pack_params(maxVal, dataLen)
    limbs = eng.pack_batch(maxVal, dataLen, [expandArray(data, dataLen)])[0]
    expected = [sum(int(w) << (64 * k) for k, w in enumerate(row)) for row in limbs]
    return {"maxInputBits": pr["maxInputBits"], "maxOutputBits": pr["numInputsPerOutput"] * pr["maxInputBits"],
            "outputSize": pr["outputSize"], "arrLen": pr["arrLen"], "expected": expected}


def unpackInput(maxVal, packedBits, data, engine=None):
    """index.js:598-620 (`data`: list of ints below 2^256)."""
    eng = engine or default_engine()
    bits = eng.pack_params(maxVal, 0)["maxInputBits"]
    per = packedBits // bits
    limbs = np.array([[(int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for v in data], dtype=np.uint64)
    un = eng.unpack_batch(maxVal, packedBits, limbs[None])[0].tolist() if len(data) else []
    return {"maxInputBits": bits, "packedBits": packedBits, "packedSize": len(data), "unpackedSize": per * len(data),
            "unpacked": trimPolynomial(un)}


# ---- the scheme class --------------------------------------------------------------------------------------

class NTRU:
    """index.js:7-207, hot-path methods.  Every return value is made of plain Python lists / ints."""

    def __init__(self, options=None, engine=None, **kw):
        opts = dict(N=167, p=3, q=128, df=61, dg=20, dr=18, f=None, fp=None, fq=None, g=None, h=None)  # index.js:9-23
        opts.update(options or {})
        opts.update(kw)
        for k, v in opts.items():
            setattr(self, k, v)
        self.I = [1] + [0] * (self.N - 1) + [-1]                                                      # index.js:25-27
        self._engine = engine

    @property
    def engine(self):
        if self._engine is None:
            self._engine = default_engine()
        return self._engine

    def calculateNq(self):
        """index.js:201-203"""
        return int(np.ceil(np.log2(float(self.q) * self.q * self.N)))

    def calculateNp(self):
        """index.js:204-206"""
        return int(np.ceil(np.log2(float(self.p) * self.p * self.N)))

    # -- encryptBits, index.js:87-110 ---------------------------------------------------------------------
    def encryptBits(self, m, r=None):
        """`r` (signed ternary, length N) may be supplied for replay; by default it is sampled like the reference."""
        N, q, p = self.N, self.q, self.p
        if r is None:
            r = generateCustomArray(N, self.dr, self.dr)
        r = [p - 1 if x == -1 else x for x in r]                                     # :89
        m_pad, h_pad = expandArray(m, N), expandArray(self.h, N)
        # addPolynomials(m, rhq, q) reduces any integer m[i] modulo q (index.js:91, :241); the device adds a byte
        m_dev = [x % q for x in m_pad]
        if q > 256 and any(x > 255 for x in m_dev):       # does not fit the byte operand: the three steps separately
            rhqm = addPolynomials(m, multiplyPolynomials(r, self.h, q, self.engine), q)
            d = dividePolynomials(rhqm, self.I, q, self.engine)
            return {"value": trimPolynomial(d["remainder"]),
                    "inputs": {"r": r, "m": m_pad, "h": h_pad, "quotientE": expandArray(d["quotient"], N + 1),
                               "remainderE": expandArray(d["remainder"], N + 1)},
                    "params": [q, self.calculateNq(), N]}
        e, quot = self.engine.encrypt_batch(N, q, h_pad, [r], [m_dev], want_quot=True)
        e, quot = e[0].tolist(), quot[0].tolist()
        return {
            "value": trimPolynomial(e),
            "inputs": {"r": r, "m": m_pad, "h": h_pad, "quotientE": quot + [0], "remainderE": e + [0]},
            "params": [q, self.calculateNq(), N],
        }

    # -- decryptBits, index.js:111-140 --------------------------------------------------------------------
    def decryptBits(self, e):
        N, q, p = self.N, self.q, self.p
        if self.f is None:
            raise TypeError("Cannot read property 'map' of null")                  # what index.js:112 does
        e_pad = expandArray(e, N)
        f_signed = expandArray(self.f, N)
        value, q1, r1, q2 = self.engine.decrypt_batch(N, q, p, f_signed, expandArray(self.fp, N), [e_pad])
        value = value[0].tolist()
        return {
            "value": trimPolynomial(value),
            "inputs": {
                "f": [q - 1 if x == -1 else x for x in f_signed],
                "fp": expandArray(self.fp, N),
                "e": e_pad,
                "quotient1": q1[0].tolist() + [0],
                "remainder1": r1[0].tolist() + [0],
                "quotient2": q2[0].tolist() + [0],
                "remainder2": value + [0],
            },
            "params": [q, self.calculateNq(), p, self.calculateNp(), N],
        }

    # -- loadPrivateKeyF, index.js:30-49 -------------------------------------------------------------------
    def loadPrivateKeyF(self, fArr):
        """fq = f^-1 mod q, fp = f^-1 mod p.  Units take the batched inversion kernels (unique inverses = the reference's,
        its validity checks pass by construction); an f those kernels flag runs the reference's own sequence on the
        generic family: same assignments in the same order, same errors, same acceptance of the non-units its `&&`
        checks let through (index.js:41-45, :451)."""
        N, q, p = self.N, self.q, self.p
        fArr = list(fArr)
        if (len(fArr) <= N and all(x in (-1, 0, 1) for x in fArr) and p == 3 and q & (q - 1) == 0 and
                self.engine.supports(N, q)):
            fq, fp, flags = self.engine.invert_key_batch(N, q, p, [expandArray(fArr, N)])
            if not int(flags[0]) & (FLAG_NOT_UNIT_MOD2 | FLAG_NOT_UNIT_MODP):
                self.f = fArr
                self.fq, self.fp = trimPolynomial(fq[0].tolist()), trimPolynomial(fp[0].tolist())
                return True
        self.f = fArr
        self.fq = polyInv(self.f, self.I, q, self.engine)
        self.fp = polyInv(self.f, self.I, p, self.engine)
        fmodq = [q - 1 if x == -1 else x for x in self.f]
        fmodp = [p - 1 if x == -1 else x for x in self.f]
        rem = dividePolynomials(multiplyPolynomials(self.fq, fmodq, q, self.engine), self.I, q, self.engine)["remainder"]
        if len(rem) != 1 and rem[0] != 1:
            raise ReferenceError_("invalid fq")
        rem = dividePolynomials(multiplyPolynomials(self.fp, fmodp, p, self.engine), self.I, p, self.engine)["remainder"]
        if len(rem) != 1 and rem[0] != 1:
            raise ReferenceError_("invalid fp")
        return True

    def generatePrivateKeyF(self, max_tries=100):
        """index.js:51-65: draw f with df ones and df - 1 minus ones until it is invertible.  Only the reference's own
        errors mean "next f"; an engine or GPU failure propagates."""
        i, retval = 0, None
        while (not retval or not (self.fq and self.fp)) and i < max_tries:
            i += 1
            try:
                retval = self.loadPrivateKeyF(generateCustomArray(self.N, self.df, self.df - 1))
            except ReferenceError_:
                pass
        if not self.fq or not self.fp:
            raise ValueError("Could not find invertible f")

    def generateNewPublicKeyGH(self):                                       # index.js:67-70
        self.g = generateCustomArray(self.N, self.dg, self.dg)
        self.generatePublicKeyH()

    # -- generatePublicKeyH, index.js:72-79 ----------------------------------------------------------------
    def generatePublicKeyH(self):
        """h = trim((p*fq mod q) * g mod (x^N - 1, q)) on the device."""
        if not self.f:
            raise ValueError("missing private key F")
        if not self.g:
            raise ValueError("missing private key G")
        if not self.fq:
            raise TypeError("Cannot read property 'map' of null")              # what index.js:76 does without fq
        h = self.engine.public_key_batch(self.N, self.q, self.p, [expandArray(self.fq, self.N)], [expandArray(self.g, self.N)])
        self.h = trimPolynomial(h[0].tolist())

    # -- verifyKeysInputs, index.js:141-197 ---------------------------------------------------------------
    def verifyKeysInputs(self):
        for attr, msg in (("f", "missing private key F"), ("fq", "missing private key Fq"),
                          ("fp", "missing private key Fp"), ("g", "missing private key G"),
                          ("h", "missing public key H")):
            if not getattr(self, attr):
                raise ValueError(msg)
        N, q, p = self.N, self.q, self.p
        pad = lambda a: expandArray(a, N)
        out = self.engine.verify_keys_batch(N, q, p, [pad(self.f)], [pad(self.g)], [pad(self.fq)], [pad(self.fp)],
                                            [pad(self.h)])
        flags = int(out["flags"][0])
        if flags & FLAG_INVALID_FQ:
            raise ValueError("invalid fq")
        if flags & FLAG_INVALID_FP:
            raise ValueError("invalid fp")
        # index.js:165 walks h as the caller stored it (untrimmed tails count): redo that literal check here
        rem_h = trimPolynomial(out["rem_h"][0].tolist())
        if flags & FLAG_INVALID_H or any(i >= len(rem_h) or rem_h[i] != cur for i, cur in enumerate(self.h)):
            raise ValueError("invalid h")
        nq, np_ = self.calculateNq(), self.calculateNp()
        row = lambda name: out[name][0].tolist() + [0]
        return {
            "fq": {"params": [q, nq, N],
                   "inputs": {"f": [q - 1 if x == -1 else x for x in pad(self.f)], "fq": pad(self.fq),
                              "quotientI": row("quot_fq"), "remainderI": row("rem_fq")}},
            "fp": {"params": [p, np_, N],
                   "inputs": {"f": [p - 1 if x == -1 else x for x in pad(self.f)], "fq": pad(self.fp),
                              "quotientI": row("quot_fp"), "remainderI": row("rem_fp")}},
            "h": {"params": [q, nq, N],
                  "inputs": {"f": [q - 1 if x == -1 else x for x in pad(self.g)], "fq": [x * p for x in pad(self.fq)],
                             "quotientI": row("quot_h"), "remainderI": row("rem_h")}},
        }

    # -- string helpers, index.js:80-86 -------------------------------------------------------------------
    # ---- additive batch API (the Node shim's ntru.pipeline): stages chained on the GPU, only m up, only what is asked for down
    def pipeline(self, m, sampleR=None, r=None, decrypt=False, pack=False, want=None):
        """m: [B][N] plaintext coefficients.  sampleR = (key[8] uint32, firstItem): r = generateCustomArray(N, dr, dr) with -1 -> p-1
        (index.js:89, :461-488) drawn on the device from a ChaCha20 stream -- or r: [B][N].  encryptBits always (index.js:87-110);
        decrypt: decryptBits of the fresh ciphertexts (index.js:111-140); pack: packOutput (index.js:572-596) of the last stage's
        result.  want: subset of {"r", "e", "value"} (default: e without decrypt, value with it, only `packed` when pack is set)."""
        N = self.N
        if (sampleR is None) == (r is None):
            raise ValueError("pipeline: give either sampleR=(key, firstItem) or r")
        want = set(want) if want is not None else (set() if pack else ({"value"} if decrypt else {"e"}))
        pad = lambda a, dt: np.array(list(a) + [0] * (N - len(a)), dtype=dt)
        key, first = (sampleR if sampleR is not None else (None, 0))
        return self.engine.pipeline_batch(N, self.q, self.p, pad(self.h, np.uint16), m,
                                          f=pad(self.f, np.int8) if decrypt else None, fp=pad(self.fp, np.uint8) if decrypt else None,
                                          key=key, first_item=first, n1=self.dr, n2=self.dr, r=r, want_r="r" in want and key is not None,
                                          want_e="e" in want, want_value="value" in want, want_packed=pack)

    def encryptStr(self, inputPlain):
        return self.encryptBits(stringToBits(inputPlain))["value"]

    def decryptStr(self, encrypted):
        bits = self.decryptBits(encrypted)["value"]
        bits = bits + [0] * (-len(bits) % 8)                                     # expandArrayToMultiple(.., 8)
        return bitsToString(bits)
