"""Host-side mirror of the reference's hot-path interface (numtel/ntru-circom index.js) over the HIP engine.

Same names, argument meaning, return shapes and error behaviour as the reference for:
  class NTRU                 index.js:7-207   (constructor options, public fields, encryptBits, decryptBits,
                                               verifyKeysInputs, encryptStr, decryptStr, calculateNq/Np)
  multiplyPolynomials        index.js:319-355
  addPolynomials             index.js:235-244 (O(N) host glue, as in the reference)
  degree / trimPolynomial / expandArray / generateCustomArray / stringToBits / bitsToString

All polynomial products and the quotient/remainder split run on the GPU through the C ABI
(include/ntru_engine.h); this file only reshapes data (padding, trimming, dict building), which is what
SURVEY.md section 8 assigns to the shim.  Key generation (index.js:30-79, 425-514) is out of scope: set
f, fp, fq, g, h through the constructor options or the public fields, exactly as the reference allows.
"""
import secrets

import numpy as np

from .engine import FLAG_INVALID_FP, FLAG_INVALID_FQ, FLAG_INVALID_H, Engine, EngineError

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


def multiplyPolynomials(a, b, p, engine=None):
    """index.js:319-355: linear product, each coefficient in [0,p), trailing zeros trimmed.

    Runs on the GPU as one polymul-split in a ring large enough to hold both operands; the linear
    product is recovered from (quotient, remainder): c[N+k] = -quot[k], c[k] = rem[k] - c[N+k]."""
    if len(a) == 0 or len(b) == 0:
        return [0]
    eng = engine or default_engine()
    N = max(len(a), len(b), 2)
    ar = np.array([x % p for x in expandArray(a, N)], dtype=np.int64)
    br = np.array([x % p for x in expandArray(b, N)], dtype=np.int64)
    quot, rem = eng.polymul_split(N, p, ar.astype(np.uint16), br.astype(np.uint16))
    hi = (p - quot[0].astype(np.int64)) % p
    lo = (rem[0].astype(np.int64) - hi) % p
    return trimPolynomial(lo.tolist() + hi.tolist()[:N - 1])


def dividePolynomials(a, b, p, engine=None):
    """index.js:358-401 for the hot path's only divisor, b = I = [1, 0, ..., 0, -1] (or mod-1 as the last entry).

    Inside encrypt/decrypt/verify the split is fused into the product kernel; a stand-alone call runs the
    elementwise split kernel (ntru_split_by_I).  Only trimming happens on the host."""
    N = len(b) - 1
    if N < 1 or b[0] % p != 1 or any(x % p != 0 for x in b[1:N]) or (b[N] % p) != p - 1:
        raise NotImplementedError("the HIP engine only divides by I = 1 - x^N (generic long division is "
                                  "key-generation code, out of scope: SURVEY.md section 8f)")
    if len(a) > 2 * N:
        raise NotImplementedError("dividend longer than 2N")
    if any(x < 0 or x >= p for x in a):
        raise NotImplementedError("dividend must already be reduced into [0, mod)")
    eng = engine or default_engine()
    quot, rem = eng.split_by_I(N, p, [expandArray(a, 2 * N)])
    nq = max(len(a) - N, 0)
    quotient = trimPolynomial(quot[0].tolist()[:nq]) if nq else [0]
    return {"quotient": quotient, "remainder": trimPolynomial(rem[0].tolist())}


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
        if any((not 0 <= x <= 255) for x in m_pad):
            raise ValueError("plaintext coefficients must be in 0..255")
        e, quot = self.engine.encrypt_batch(N, q, h_pad, [r], [m_pad], want_quot=True)
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
        """fq = f^-1 mod q, fp = f^-1 mod p on the device.  For f that is not a unit this raises 'invalid_gcd' (the
        reference throws that or 'invalid fq' for most such f and accepts a few through its `&&` checks)."""
        from .engine import FLAG_NOT_UNIT_MOD2, FLAG_NOT_UNIT_MODP
        fq, fp, flags = self.engine.invert_key_batch(self.N, self.q, self.p, [expandArray(list(fArr), self.N)])
        if int(flags[0]) & (FLAG_NOT_UNIT_MOD2 | FLAG_NOT_UNIT_MODP):
            raise ValueError("invalid_gcd")
        self.f = list(fArr)
        self.fq, self.fp = trimPolynomial(fq[0].tolist()), trimPolynomial(fp[0].tolist())
        return True

    def generatePrivateKeyF(self, max_tries=100):
        """index.js:51-65: draw f with df ones and df - 1 minus ones until it is invertible."""
        for _ in range(max_tries):
            try:
                return self.loadPrivateKeyF(generateCustomArray(self.N, self.df, self.df - 1))
            except ValueError:
                continue
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
    def encryptStr(self, inputPlain):
        return self.encryptBits(stringToBits(inputPlain))["value"]

    def decryptStr(self, encrypted):
        bits = self.decryptBits(encrypted)["value"]
        bits = bits + [0] * (-len(bits) % 8)                                     # expandArrayToMultiple(.., 8)
        return bitsToString(bits)
