#!/usr/bin/env python3
"""Executable specification of the int8 matrix-core path (kernel family 4, `k_*_m` in csrc/matrix_encrypt.hip / matrix_decrypt.hip / matrix_common.h).

A product of a batch operand X[b][i] with a SHARED key operand s is a matrix product with the Toeplitz matrix of s:
    low [b][k] = sum_{i <= k} X[b][i] s[k-i]            (coefficients 0..N-1 of the linear product)
    high[b][k] = sum_{i >  k} X[b][i] s[k-i+N]          (coefficients N..2N-1)
    remainder = low + high,  quotient = -high            (division by 1 - x^N in closed form)
Tiles are 32 x 32 x 32 (v_mfma_i32_32x32x32_i8): rows = items, columns = output coefficients k = 32 kb + k',
contraction i = 32 ib + i'.  The tile of the Toeplitz matrix depends only on d = kb - ib:
    G_d[i'][k'] = sc[32 d + k' - i'],  sc = s extended with period N
d > 0 feeds low, d < 0 feeds high, d = 0 is split by k' >= i'.  Operands wider than int8 use two digit planes on the
contraction axis: value = lo + 128 * hi, computed as [A | alpha*A] x [lo ; beta*hi] with alpha*beta = 128.

This model reproduces the fragment addressing (reversed cyclic byte array, 4 byte-shifted copies), the digit planes,
the diagonal masks and the wave strips with numpy, and checks encrypt / decrypt against a direct convolution.
"""
import numpy as np


def tiles(N):
    NT = (N + 31) // 32
    return NT, 32 * NT


def signed_rep(h, q):
    """Representative of h mod q whose digits fit: d0 in [-64, 63], 4*d1 in [-128, 124] (q <= 8192)."""
    h = np.asarray(h, dtype=np.int64) % q
    hs = np.where(h > 4031, h - q, h) if q == 8192 else np.where(h >= q // 2, h - q, h)
    d0 = ((hs + 64) & 127) - 64
    d1 = (hs - d0) >> 7
    assert np.all(d0 + 128 * d1 == hs) and np.all(np.abs(d0) <= 64) and d1.min() >= -32 and d1.max() <= 31
    return d0, d1


def rev_array(plane, N):
    """rev[y] = plane[(Y0 - y) mod N], Y0 = 32 NT - 1, length 64 NT."""
    NT, _ = tiles(N)
    y = np.arange(64 * NT)
    return plane[(32 * NT - 1 - y) % N].astype(np.int8)


def b_fragment(rev, N, d):
    """[64 lanes][16] bytes: lane (r = l & 31, h = l >> 5), element j = G_d[i' = 16 h + j][k' = r]."""
    NT, _ = tiles(N)
    lane = np.arange(64)
    r, h = lane & 31, lane >> 5
    y0 = 32 * NT - 1 - 32 * d - r + 16 * h
    assert y0.min() >= 0 and y0.max() + 15 < 64 * NT
    copy, dw = y0 & 3, y0 >> 2                      # the kernel reads 4 dwords of copy `y0 & 3` at dword y0 >> 2
    frag = np.stack([rev[4 * dw + copy + j] for j in range(16)], axis=1)
    return frag


def diag_masks():
    lane = np.arange(64)
    r, h = lane & 31, lane >> 5
    ip = 16 * h[:, None] + np.arange(16)[None, :]
    low = (r[:, None] >= ip)
    return low, ~low


def a_fragment(stage, ib):
    """stage [32 rows][NP] int8 -> [64 lanes][16]: lane (r, h) holds X[row r][32 ib + 16 h + j]."""
    lane = np.arange(64)
    r, h = lane & 31, lane >> 5
    cols = 32 * ib + 16 * h[:, None] + np.arange(16)[None, :]
    return stage[r[:, None], cols]


def mfma(a, b, acc):
    """v_mfma_i32_32x32x32_i8: D[row][col] += sum_kk A[row][kk] B[kk][col] with both operands in lane-fragment form."""
    A = np.zeros((32, 32), np.int64); Bm = np.zeros((32, 32), np.int64)
    lane = np.arange(64)
    r, h = lane & 31, lane >> 5
    for j in range(16):
        A[r, 16 * h + j] = a[:, j]
        Bm[16 * h + j, r] = b[:, j]
    return acc + A @ Bm


def strips(NT, waves=4):
    """4 R strips of <= 4 tiles in column order; in round rho wave w takes strip 4 rho + w (adjacent strips are produced
    at the same time).  Returns per wave the list of (kb0, nt)."""
    rounds = -(-(-(-NT // 4)) // waves)
    n_str = rounds * waves
    base, rem = NT // n_str, NT % n_str
    out = [[] for _ in range(waves)]
    for j in range(n_str):
        nt = base + (1 if j < rem else 0)
        if nt:
            out[j % waves].append((j * base + min(j, rem), nt))
    assert sum(nt for w in out for _, nt in w) == NT and max(nt for w in out for _, nt in w) <= 4
    return out


def toeplitz_product(a_planes, b_planes_rev, N):
    """a_planes: list of [32][NP] int8 stages; b_planes_rev: list of reversed arrays (same length).  Returns
    (low, high) [32][NP] int64 following the kernel's loop structure."""
    NT, NP = tiles(N)
    low = np.zeros((32, NP), np.int64); high = np.zeros((32, NP), np.int64)
    mlow, mhigh = diag_masks()
    for wave in strips(NT):
        for kb0, nt in wave:
            accL = [np.zeros((32, 32), np.int64) for _ in range(nt)]
            accH = [np.zeros((32, 32), np.int64) for _ in range(nt)]
            for ib in range(NT):
                for a_st, rev in zip(a_planes, b_planes_rev):
                    a = a_fragment(a_st, ib)
                    for t in range(nt):
                        d = kb0 + t - ib
                        g = b_fragment(rev, N, d)
                        if d > 0:
                            accL[t] = mfma(a, g, accL[t])
                        elif d < 0:
                            accH[t] = mfma(a, g, accH[t])
                        else:
                            accL[t] = mfma(a, np.where(mlow, g, 0), accL[t])
                            accH[t] = mfma(a, np.where(mhigh, g, 0), accH[t])
            for t in range(nt):
                low[:, 32 * (kb0 + t):32 * (kb0 + t + 1)] = accL[t]
                high[:, 32 * (kb0 + t):32 * (kb0 + t + 1)] = accH[t]
    return low, high


def stage(rows, N, dtype=np.int8):
    NT, NP = tiles(N)
    st = np.zeros((32, NP), dtype)
    st[:, :N] = rows
    return st


def encrypt_model(N, q, h, r, m):
    d0, d1 = signed_rep(h, q)
    X = stage(r, N)
    low, high = toeplitz_product([X, (X.astype(np.int64) << 5).astype(np.int8)],
                                 [rev_array(d0, N), rev_array(4 * d1, N)], N)
    rem = (low[:, :N] + high[:, :N] + m) % q
    quot = (-high[:, :N]) % q
    return rem, quot


def lift(x, q, p):
    return np.where(2 * x > q, (x + 1) % p, x % p)


def decrypt_model(N, q, p, f, fp, e):
    e = np.asarray(e, np.int64)
    E_lo, E_hi2 = stage(e & 127, N), stage((e >> 6) & 0xFE, N)
    fs = np.asarray(f, np.int64)
    low, high = toeplitz_product([E_lo, E_hi2], [rev_array(fs, N), rev_array(64 * fs, N)], N)
    rem1 = (low[:, :N] + high[:, :N]) % q
    quot1 = (-high[:, :N]) % q
    bl = lift(rem1, q, p)
    low2, high2 = toeplitz_product([stage(bl, N)], [rev_array(np.asarray(fp, np.int64), N)], N)
    value = (low2[:, :N] + high2[:, :N]) % p
    quot2 = (-high2[:, :N]) % p
    return value, quot1, rem1, quot2


def direct_split(X, s, N, mod):
    lin = np.zeros((X.shape[0], 2 * N), np.int64)
    for i in range(N):
        lin[:, i:i + N] += X[:, i:i + 1] * s[None, :]
    return (lin[:, :N] + lin[:, N:]) % mod, (-lin[:, N:]) % mod


def main():
    rng = np.random.default_rng(5)
    for N, q in ((821, 4096), (701, 8192), (167, 128), (509, 2048), (33, 32), (64, 8192)):
        p = 3
        h = rng.integers(0, q, N)
        if q == 8192:
            h[:4] = (4031, 4032, 4095, 4096)               # the corner of the digit range
        f = rng.integers(-1, 2, N); fp = rng.integers(0, 3, N)
        r = rng.integers(0, 3, (32, N)); m = rng.integers(0, 256, (32, N))
        rem, quot = encrypt_model(N, q, h, r, m)
        rr, qq = direct_split(r, h, N, q)
        assert np.array_equal(rem, (rr + m) % q) and np.array_equal(quot, qq), (N, q, "encrypt")
        e = rng.integers(0, q, (32, N))
        value, quot1, rem1, quot2 = decrypt_model(N, q, p, f, fp, e)
        r1, q1 = direct_split(e, f % q, N, q)
        assert np.array_equal(rem1, r1) and np.array_equal(quot1, q1), (N, q, "decrypt product 1")
        v2, q2 = direct_split(lift(r1, q, p), fp, N, p)
        assert np.array_equal(value, v2) and np.array_equal(quot2, q2), (N, q, "decrypt product 2")
        print("N=%d q=%d: tiles %d, strips %s  OK" % (N, q, tiles(N)[0], strips(tiles(N)[0])))


if __name__ == "__main__":
    main()
