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


def product_split_registers(a_planes, s, N):
    """The form the kernels use since round 5 (csrc/matrix_peritem.hip, "chunk rows in REGISTERS"): lane (r, hh) of the wave holds
    bytes 16 hh .. 16 hh + 15 of chunk r of every plane; the low part walks d = 1, 2, ... by moving every lane's 16 bytes one lane UP
    (wave_shr:1, lane 0 takes zero, lane 32 -- the seam between the half-waves -- is cut to zero), the high part walks d = -1, -2, ...
    one lane DOWN (wave_shl:1, lanes 63 and 31 take zero).  No row of the chunk matrix is ever read from the LDS again; the fragment of
    distance d is read once for ALL planes.  Returns (low[p], high[p]) per plane and the number of fragment reads."""
    NT = tiles(N)
    lowm = np.arange(32)[None, :] >= np.arange(32)[:, None]
    lanes = []                                                          # [plane][64 lanes][16 bytes]
    for a in a_planes:
        ap = np.zeros(32 * 32, np.int64); ap[:N] = a                    # chunks >= NT are zero: lanes r >= NT hold zeros
        lanes.append(np.stack([ap[32 * r + 16 * hh: 32 * r + 16 * hh + 16] for hh in (0, 1) for r in range(32)]))
    def as_matrix(L):                                                   # the MFMA A operand of these 64 lanes: rows r, K = 16 hh + j
        return np.concatenate([L[:32], L[32:]], axis=1)
    def up(L):
        M = np.zeros_like(L); M[1:] = L[:-1]; M[32] = 0; return M      # wave_shr:1 + seam mask (lane 32)
    def down(L):
        M = np.zeros_like(L); M[:-1] = L[1:]; M[31] = 0; return M      # wave_shl:1 + seam mask (lane 31)
    out, reads = [], 0
    G0 = toeplitz_tile(s, N, 0); reads += 1
    accL = [as_matrix(L) @ np.where(lowm, G0, 0) for L in lanes]
    accH = [as_matrix(L) @ np.where(~lowm, G0, 0) for L in lanes]
    AL, AH = [L.copy() for L in lanes], [L.copy() for L in lanes]
    for j in range(1, NT):
        Gl, Gh = toeplitz_tile(s, N, j), toeplitz_tile(s, N, -j); reads += 2
        for p in range(len(lanes)):
            AL[p] = up(AL[p]); accL[p] = accL[p] + as_matrix(AL[p]) @ Gl
            AH[p] = down(AH[p]); accH[p] = accH[p] + as_matrix(AH[p]) @ Gh
    for p in range(len(lanes)):
        out.append((accL[p].reshape(-1)[:N], accH[p].reshape(-1)[:N]))
    return out, reads


def tile_shapes(N):
    """Go / no-go on paper for the 16-row tile (v_mfma_i32_16x16x64_i8) against the 32-row one, per digit plane and product:
    matrix work issued, the useful share of it, operand bytes an instruction needs and lane shifts per plane when the rows live
    in registers."""
    NT = tiles(N)
    useful = NT * NT                                                    # tile pairs (kb, ib), 32 x 32 x 32 MACs each
    issued32 = 2 * NT * 32                                              # 2 NT instructions x 32 rows
    # 16-row tile: row groups of 16 output tiles; a (group, d) unit is issued when any of its rows is in range
    units = 0
    for part, ds in (("low", range(0, NT)), ("high", range(-(NT - 1), 0))):
        for g0 in range(0, NT, 16):
            rows = range(g0, min(g0 + 16, NT))
            units += sum(1 for d in ds if any(0 <= kb - d < NT for kb in rows))
    instr16 = 2 * ((units + 1) // 2)                                    # K = 64: two distances per instruction, two 16-column halves
    print("N=%d (NT=%d), per plane and product:" % (N, NT))
    print("  32x32x32: %3d instructions (%5d clocks), useful MACs %.0f %%; operand bytes per instruction: 1 KB fragment (+ 1 KB of rows when "
          "they are read from the LDS); 4 lane shifts per distance with the rows in registers" % (2 * NT, 2 * NT * 32, 100.0 * useful / issued32))
    print("  16x16x64: %3d instructions (%5d clocks), useful MACs %.0f %%; operand bytes per instruction: 1 KB fragment per (distance pair, "
          "column half) shared by the row groups; 4 + 8 lane shifts per distance PAIR and plane (two row groups, the second takes its entering rows "
          "from the first: 6 per distance)" % (instr16, instr16 * 16, 100.0 * useful / (units * 16.0)))
    print("  -> matrix clocks %.2fx, lane shifts 1.5x: with the vector port as loaded as the matrix pipe (round 5 counters: 0.9 against 0.67) the "
          "16-row tile is not built" % (instr16 * 16 / (2 * NT * 32.0)))


def main():
    tile_shapes(821)
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
        # rows in registers: operands swapped (the 13-bit operand's planes are the rows, the ternary one the Toeplitz fragments), as
        # k_verify_keys_m runs products 1 and 2: planes fq lo, fq hi, fp against the fragments of f -- ONE fragment read per distance
        v = fq % q
        planes, reads = product_split_registers([v & 127, v >> 7, fp], f, N)
        lo = planes[0][0] + 128 * planes[1][0]; hi = planes[0][1] + 128 * planes[1][1]
        assert np.array_equal((lo + hi) % q, (lin[:N] + lin[N:]) % q) and np.array_equal((-hi) % q, (-lin[N:]) % q), (N, q, "registers")
        assert np.array_equal(planes[2][0], low3) and np.array_equal(planes[2][1], high3), (N, "registers, mod p")
        print("N=%d q=%d: %d + %d matrix instructions per item for (f * fq, f * fp); rows in registers: %d fragment reads for the three "
              "planes and no row reads (rows in the LDS: %d fragment + %d row reads); packed-MAC wave instructions of the vector-ALU family: %d  OK"
              % (N, q, n, n3, reads, 2 * reads, 3 * reads, 2 * ((N * N + 127) // 128)))


if __name__ == "__main__":
    main()
