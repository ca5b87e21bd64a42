#!/usr/bin/env python3
"""Executable specification of a per-item product on the int8 matrix cores (`k_verify_keys_m`, `k_product_tern_m`,
`k_polymul_m` in csrc/matrix_peritem.hip).

Family 4 (tools/mfma_model.py) needs a key shared by the batch.  A product with per-item operands (verifyKeysInputs,
index.js:141-197; the Newton rounds of polyInv, index.js:491-514) has no shared matrix, but one product c = a * s in
Z[x] is itself a 32-row matrix product per tile distance:
    i = 32 ib + i',  k = 32 kb + k',  d = kb - ib in (-NT, NT)
    C[kb][k'] += sum_{i'} F[kb - d][i'] * G_d[i'][k'],     F[ib][i'] = a[32 ib + i'],  G_d[i'][k'] = sc[32 d + k' - i']
so for a fixed d ONE v_mfma_i32_32x32x32_i8 (rows = kb, columns = k', contraction = i') adds the contribution of every
tile pair at that distance; the A operand is the chunk matrix of `a` shifted down by d rows (zero rows outside), the B
operand is the same Toeplitz fragment family 4 reads from the reversed cyclic array of s.  d > 0 feeds the low half
(coefficients 0..N-1), d < 0 the high half (N..2N-1), d = 0 is split by k' >= i'.  One accumulator pair holds the whole
item: 2 NT - 1 matrix instructions per digit plane.  An operand wider than int8 uses digit planes as in family 4.

This model reproduces that loop structure with numpy and checks it against a direct convolution for the verify_keys
operand shapes (ternary x 13-bit, ternary x 2-bit)."""
import numpy as np


def tiles(N):
    NT = (N + 31) // 32
    assert NT <= 32, "one 32-row accumulator holds at most 32 output tiles (N <= 1024)"
    return NT


def chunk_rows(a, N, d):
    """A operand for tile distance d: row kb = chunk kb - d of `a` (32 coefficients), zero outside [0, NT)."""
    NT = tiles(N)
    ap = np.zeros(32 * NT, np.int64); ap[:N] = a
    F = ap.reshape(NT, 32)
    A = np.zeros((32, 32), np.int64)
    for kb in range(NT):
        ib = kb - d
        if 0 <= ib < NT:
            A[kb] = F[ib]
    return A


def toeplitz_tile(s, N, d):
    """G_d[i'][k'] = sc[32 d + k' - i'] with sc = s extended with period N (the cyclic array of family 4)."""
    ip, kp = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    return np.asarray(s, np.int64)[(32 * d + kp - ip) % N]


def digits(s, q):
    """13-bit operand -> two int8 planes with s = d0 + 128 d1 (mod q): the signed representative of family 4."""
    s = np.asarray(s, np.int64) % q
    hs = np.where(s > q // 2 - 65, s - q, s) if q > 128 else s
    d0 = ((hs + 64) & 127) - 64
    d1 = (hs - d0) >> 7
    assert np.all(np.abs(d0) <= 64) and np.all(np.abs(4 * d1) <= 128) and np.all((d0 + 128 * d1 - s) % q == 0)
    return d0, d1


def product_split(a, s_planes, scales, N):
    """a: small operand (int8 range); s_planes: digit planes of the other operand, value = sum scale_p * plane_p.
    Returns (low, high) [N] int64 = coefficients 0..N-1 and N..2N-1 of the linear product a * s."""
    NT = tiles(N)
    lowm = np.arange(32)[None, :] >= np.arange(32)[:, None]            # [i'][k']: k' >= i' -> low half
    accL = np.zeros((32, 32), np.int64); accH = np.zeros((32, 32), np.int64)
    n_mfma = 0
    for plane, scale in zip(s_planes, scales):
        assert np.abs(plane).max() * 1 <= 128
        for d in range(-(NT - 1), NT):
            A = chunk_rows(a, N, d) * scale                            # the scale rides on the small operand (|a| * scale <= 127)
            assert np.abs(A).max() <= 127
            G = toeplitz_tile(plane, N, d)
            if d > 0:
                accL += A @ G; n_mfma += 1
            elif d < 0:
                accH += A @ G; n_mfma += 1
            else:
                accL += A @ np.where(lowm, G, 0); accH += A @ np.where(~lowm, G, 0); n_mfma += 2
    return accL.reshape(-1)[:N], accH.reshape(-1)[:N], n_mfma


def main():
    rng = np.random.default_rng(11)
    for N, q in ((821, 4096), (701, 8192), (509, 2048), (167, 128), (1024, 8192), (33, 32)):
        f = rng.integers(-1, 2, N)                                      # ternary operand
        fq = rng.integers(0, q, N)                                      # 13-bit operand
        lin = np.convolve(f, fq)
        lin = np.concatenate([lin, np.zeros(2 * N - len(lin), np.int64)])
        d0, d1 = digits(fq, q)
        # value = d0 + 128 d1 = d0 * 1 + (4 d1) * 32: planes [d0 ; 4 d1] against [f | 32 f]
        low, high, n = product_split(f, [d0, 4 * d1], [1, 32], N)
        assert np.array_equal((low + high) % q, (lin[:N] + lin[N:]) % q) and np.array_equal((-high) % q, (-lin[N:]) % q), (N, q)
        fp = rng.integers(0, 3, N)
        low3, high3, n3 = product_split(f, [fp], [1], N)
        lin3 = np.convolve(f, fp); lin3 = np.concatenate([lin3, np.zeros(2 * N - len(lin3), np.int64)])
        assert np.array_equal(low3, lin3[:N]) and np.array_equal(high3, lin3[N:]), (N, "mod p")
        print("N=%d q=%d: %d + %d matrix instructions per item for (f * fq, f * fp); packed-MAC wave instructions today: %d  OK"
              % (N, q, n, n3, 2 * ((N * N + 127) // 128)))


if __name__ == "__main__":
    main()
