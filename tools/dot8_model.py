"""Numpy model of the v_dot8_u32_u4 formulation of the ternary x ternary product (design check, not product code).

Item layout: lane l (0..LANES-1) owns outputs k = 32*l + t, t in [0,32).  A8r[I] packs fp[8I+7-i] in nibble i (shared).
The per-item operand b lives in a nibble stream: coefficient j (cyclic, j in [-ORG, ...)) at nibble index j + ORG.
FB(t, I) = the 8 nibbles b[k-8I-7 .. k-8I] (forward) = funnel(dw[w+1], dw[w], 4*ph).  acc[t] += dot8(A8r[I], FB(t, I)).
"""
import numpy as np



def nib(word, i):
    return (word >> (4 * i)) & 15


def dot8(a, b):
    return sum(nib(a, i) * nib(b, i) for i in range(8))


def funnel(hi, lo, sh):
    return ((((hi << 32) | lo) >> sh) & 0xFFFFFFFF) if sh else lo


def build_stream(b, N, ndw, ORG):
    """dw[] as the kernel builds it: positive part from bytes, negative part via funnel shifts of the positive part."""
    nbytes = ndw * 4
    by = np.zeros(nbytes, np.int64)
    for c in range((N + 1) // 2):
        lo = int(b[2 * c]); hi = int(b[2 * c + 1]) if 2 * c + 1 < N else 0
        by[ORG // 2 + c] = lo | (hi << 4)
    dw = [int(by[4 * w]) | int(by[4 * w + 1]) << 8 | int(by[4 * w + 2]) << 16 | int(by[4 * w + 3]) << 24 for w in range(ndw)]
    return dw


def model(fp, b, N, first_neg_dw=0):
    LANES = (N + 31) // 32
    NBLK = 4 * LANES
    ORG = 8 * NBLK                     # nibble index of coefficient 0
    ndw = 4 * LANES + NBLK + 12
    dw = build_stream(b, N, ndw, ORG)
    a, r4 = N >> 3, 4 * (N & 7)        # coefficient j < 0 is b[j+N]: N nibbles = a dwords + r nibbles further up
    for w in range(first_neg_dw, NBLK):
        dw[w] = funnel(dw[w + a + 1], dw[w + a], r4) if r4 else dw[w + a]
    for n in range(ORG - (N - 1), ORG + N):        # coefficients -(N-1) .. N-1: everything a valid output can read
        j = n - ORG
        assert nib(dw[n >> 3], n & 7) == int(b[j % N]), (n, j)
    fpp = list(fp) + [0] * (8 * NBLK + 8 - N)
    A8r = [sum(int(fpp[8 * I + 7 - i]) << (4 * i) for i in range(8)) for I in range(NBLK)]
    acc = np.zeros((LANES, 32), np.int64); snap = np.zeros((LANES, 32), np.int64)
    S = np.zeros((LANES, 32), np.int64)
    Y = np.zeros(LANES, np.int64)
    for l in range(LANES):            # initial window (block 0)
        D = [dw[4 * l + NBLK - 1 + g] for g in range(6)]
        for t in range(32):
            g, ph = (t + 1) >> 3, (t + 1) & 7
            S[l, t & 31] = funnel(D[g + 1], D[g], 4 * ph) if ph else D[g]
        Y[l] = D[0]
    for I in range(NBLK):
        for l in range(LANES):
            if I == 4 * l: snap[l] = acc[l]
            for t in range(32):
                acc[l, t] += dot8(A8r[I], int(S[l, (t - 8 * I) & 31]))
            X = dw[4 * l - I + NBLK - 2] if 4 * l - I + NBLK - 2 >= 0 else 0
            for tp in range(7):
                S[l, (tp - 8 * (I + 1)) & 31] = funnel(int(Y[l]), X, 4 * (tp + 1))
            S[l, (7 - 8 * (I + 1)) & 31] = int(Y[l])
            Y[l] = X
    # correction: in-block part of the low half
    ZD = [0, 0, 0, 0] + [dw[NBLK + g] for g in range(5)]
    low = snap.copy()
    for l in range(LANES):
        for d in range(4):
            a = A8r[4 * l + d]
            for t in range(32):
                g, ph = ((t + 1) >> 3) + 3 - d, (t + 1) & 7
                low[l, t] += dot8(a, funnel(ZD[g + 1], ZD[g], 4 * ph) if ph else ZD[g])
    T = acc.reshape(-1)[:N]; lo = low.reshape(-1)[:N]
    return T, lo


if __name__ == "__main__":
    rng = np.random.default_rng(3)
    for N in (821, 701, 677, 509, 449):
        fp = rng.integers(0, 3, N); b = rng.integers(0, 3, N)
        T, lo = model(fp, b, N)
        c = np.convolve(fp, b)                     # linear product, length 2N-1
        cyc = c[:N].copy(); cyc[:N - 1] += c[N:]
        assert (T == cyc).all(), "cyclic product mismatch"
        assert (lo == c[:N]).all(), "low half mismatch"
        print("dot8 model ok: T, low match the direct convolution for N=%d" % N)
    try:
        model(rng.integers(0, 3, 701), rng.integers(0, 3, 701), 701, first_neg_dw=1)
        print("N=701 also fine when dword 0 is skipped")
    except AssertionError as e:
        print("N=701 breaks when the negative part starts at dword 1:", e)
