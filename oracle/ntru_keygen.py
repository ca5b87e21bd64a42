"""TEST INFRASTRUCTURE ONLY (see oracle/ntru_oracle.c): CPU restatement of the reference's key inversion,
SURVEY.md 8(f) #1.  Nothing under ntru-circom_amd/ may import this.

Follows index.js:
  modInverse / subtractPolynomials / multiplyPolynomialsByScalar / multiplyPolynomials (exact) / dividePolynomials
                              :224-232, :247-256, :404-406, :319-355, :358-401  -- pinned one by one by
                              tests/test_oracle_golden.py against tests/golden/generic_functions.json
  extendedEuclideanAlgorithm  :425-459   (with dividePolynomials :358-401, subtractPolynomials :247-257)
  polyInv                     :491-514   (EEA mod 2 + `exponent - 1` rounds of v <- 2v - f v^2 for a power of two,
                                          plain EEA for a prime)
  loadPrivateKeyF             :30-49     (fq = polyInv(f, I, q), fp = polyInv(f, I, p), then the `&&` validity check)
Pinned by tests/test_oracle_golden.py::test_key_inversion_equals_reference_keys against every captured key (f -> fq, fp)
and against the captured failing / quirky cases of tests/golden/keygen_cases.json.
Plain numpy, O(N^2) per key: a 821-coefficient key takes a fraction of a second."""
import numpy as np


class InvalidGcd(Exception):
    """extendedEuclideanAlgorithm's `throw new Error('invalid_gcd')` (index.js:452)."""


def _deg(a):
    nz = np.nonzero(a)[0]
    return int(nz[-1]) if nz.size else -1


def _trim(a):
    d = _deg(a)
    return a[:d + 1].copy() if d >= 0 else np.zeros(1, np.int64)


def _mod_inverse(a, p):
    """index.js:224-232: the smallest x in 1..p-1 with (a * x) % p == 1 on the normalised residue, None if there is
    none.  The reference searches x upwards; an inverse modulo p is unique in that range, so for large p (2^20) the
    search is replaced by the integer Euclidean algorithm -- same value, pinned by the modInverse vectors."""
    a = ((int(a) % p) + p) % p
    if p <= 4096:
        for x in range(1, p):
            if (a * x) % p == 1:
                return x
        return None
    import math
    return pow(a, -1, p) if a and math.gcd(a, p) == 1 else None


def _divide(a, b, p):
    """dividePolynomials (index.js:358-401); coefficients of `a` below the divisor's degree are NOT reduced."""
    if _deg(b) == -1:
        raise ZeroDivisionError("Cannot divide by zero polynomial.")
    dividend = np.array(a, dtype=np.int64)
    divisor = np.array(b, dtype=np.int64)
    dd = _deg(divisor)
    quotient = np.zeros(max(0, _deg(dividend) - dd + 1), np.int64)
    while _deg(dividend) >= dd:
        dg = _deg(dividend)
        inv = _mod_inverse(divisor[dd], p)
        if inv is None:
            raise ArithmeticError("No inverse exists for division.")
        coeff = int(np.fmod(dividend[dg] * inv, p))            # JS %: sign of the dividend
        diff = dg - dd
        quotient[diff] = coeff
        seg = np.fmod(dividend[diff:diff + dd + 1] - coeff * divisor[:dd + 1], p)
        seg[seg < 0] += p
        dividend[diff:diff + dd + 1] = seg
    return _trim(quotient), _trim(dividend)


def _multiply(a, b, p):
    """multiplyPolynomials (index.js:319-355): exact linear product, every coefficient into [0, p), trimmed."""
    if len(a) == 0 or len(b) == 0:
        return np.zeros(1, np.int64)
    return _trim(np.convolve(np.asarray(a, np.int64), np.asarray(b, np.int64)) % p)


def _subtract(a, b, p):
    n = max(len(a), len(b))
    x = np.zeros(n, np.int64); y = np.zeros(n, np.int64)
    x[:len(a)] = a; y[:len(b)] = b
    return _trim((np.fmod(x - y, p) + p) % p)


def extended_euclid(a, b, p, want_gcd=False):
    """extendedEuclideanAlgorithm(a, b, p) -> inverse of a modulo b (index.js:425-459), quirks included;
    (gcd, inverse) with want_gcd."""
    r0, r1 = np.array(a, dtype=np.int64), np.array(b, dtype=np.int64)
    s0, s1 = np.array([1], np.int64), np.array([0], np.int64)
    while _deg(r1) >= 0:
        quotient, remainder = _divide(r0, r1, p)
        r0, r1 = r1, remainder
        s0, s1 = s1, _subtract(s0, _multiply(quotient, s1, p), p)
    inv_lead = _mod_inverse(r0[_deg(r0)], p) if _deg(r0) >= 0 else None      # modInverse(undefined) is null

    if inv_lead is not None and inv_lead != 1:
        r0 = np.fmod(r0 * inv_lead, p)
        s0 = np.fmod(s0 * inv_lead, p)
    if len(r0) != 1 and (len(r0) == 0 or r0[0] != 1):            # the reference's `&&` (index.js:451)
        raise InvalidGcd("invalid_gcd")
    return (r0, s0) if want_gcd else s0


def scale(poly, s, p):
    """multiplyPolynomialsByScalar (index.js:404-406): JS `%`, no normalisation, no trimming."""
    return np.fmod(np.asarray(poly, np.int64) * s, p)


def poly_inv_generic(f, poly_i, mod):
    """polyInv(f, polyI, mod) for ANY modulus polynomial (index.js:491-514)."""
    f = np.asarray(f, np.int64)
    I = np.asarray(poly_i, np.int64)
    e = np.log2(mod)
    if round(e) == e:
        inverse = extended_euclid(f, I, 2)
        for _ in range(1, int(e)):
            twice = np.fmod(inverse * 2, mod)
            cube = _multiply(f, _multiply(inverse, inverse, mod), mod)
            upd = _subtract(twice, cube, mod)
            _, rem = _divide(upd, I, mod)
            inverse = _trim(rem)
        return inverse
    return extended_euclid(f, I, mod)


def poly_inv(f, N, mod):
    """polyInv(f, I, mod) with I = 1 - x^N as the reference builds it (index.js:25-27): [1, 0, ..., 0, -1]."""
    I = np.zeros(N + 1, np.int64); I[0] = 1; I[N] = -1
    return poly_inv_generic(f, I, mod)


def load_private_key(f, N, q, p):
    """loadPrivateKeyF (index.js:30-49): returns (fq, fp) padded to N, or raises what the reference throws."""
    f = np.asarray(f, np.int64)
    fq = poly_inv(f, N, q)
    fp = poly_inv(f, N, p)

    def check(inv, mod, msg):
        fm = np.where(f == -1, mod - 1, f)
        I = np.zeros(N + 1, np.int64); I[0] = 1; I[N] = -1
        _, rem = _divide(_multiply(inv, fm, mod), I, mod)
        if len(rem) != 1 and rem[0] != 1:
            raise ValueError(msg)
    check(fq, q, "invalid fq")
    check(fp, p, "invalid fp")
    pad = lambda a: np.concatenate([a, np.zeros(N - len(a), np.int64)]) if len(a) < N else a[:N]
    return pad(fq), pad(fp)


def is_unit(f, N, p):
    """True iff f is invertible in Z_p[x]/(x^N - 1) (p prime): gcd(f, x^N - 1) = 1 by the plain Euclidean algorithm."""
    r0 = np.zeros(N + 1, np.int64); r0[0] = p - 1; r0[N] = 1
    r1 = _trim(np.asarray(f, np.int64) % p)
    while _deg(r1) >= 0:
        _, rem = _divide(r0, r1, p)
        r0, r1 = r1, rem
    return _deg(r0) == 0
