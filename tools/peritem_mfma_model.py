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
operand shapes (ternary x 13-bit, ternary x 2-bit); `product_split_registers` replays the lane shifts of the rows-in-registers
kernels, `product_cyclic_registers` the one-instruction-per-distance form of the products modulo x^N - 1 (Newton rounds, public key)."""
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


def product_cyclic_registers(a_planes, s, N):
    """Products whose QUOTIENT nobody wants (the Newton rounds of polyInv, the public key h): c = a * s modulo x^N - 1 with ONE matrix
    instruction per tile distance d = 0 .. NT - 1 and ONE accumulator (`pi_product_cyc`, csrc/matrix_peritem.hip).  The wrapped terms
    a[i] s[k - i + N] (i > k) have the SAME Toeplitz fragment as the unwrapped ones of distance d when their rows come from a copy of
    `a` moved up by P = 32 NT - N places (a2[i + P] = a[i]): with j = i + P = 32 (kb - d + NT) + j1 (chunk kb - d + NT of a2),
        k - i + N = 32 kb + k1 - 32 (kb - d + NT) - j1 + P + N = 32 d + k1 - j1.
    So row kb of distance d holds chunk kb - d of a (kb >= d) or chunk kb - d + NT of a2 (kb < d): the rows walk UP one lane per
    distance and chunk NT - d of a2 ENTERS at row 0 (two lanes, from a byte image of a2 in the LDS, by v_cndmask_b32_dpp).  The tile
    of distance 0 is taken whole (its entries above the diagonal are the wrapped terms inside the tile), so the P coefficients that
    chunk kb + 1 of a2 shares with chunk kb of a are cut from the rows of the LAST distance, where that chunk sits at row kb.
    Returns the products modulo x^N - 1 per plane and the number of matrix instructions per plane."""
    NT = tiles(N)
    P = 32 * NT - N
    lanes, images = [], []
    for a in a_planes:
        ap = np.zeros(32 * 32, np.int64); ap[:N] = a
        lanes.append(np.stack([ap[32 * r + 16 * hh: 32 * r + 16 * hh + 16] for hh in (0, 1) for r in range(32)]))
        img = np.zeros(32 * NT, np.int64); img[P:] = a                   # the LDS image: byte P + i = a[i]
        images.append(img)
    def as_matrix(L):
        return np.concatenate([L[:32], L[32:]], axis=1)
    def up_enter(L, e):                                                 # wave_shr:1; lanes 0 and 32 (row 0) take the entering row's halves
        M = np.zeros_like(L); M[1:] = L[:-1]
        M[0] = e[:16]; M[32] = e[16:]
        return M
    cut = np.zeros((64, 16), bool)                                      # last distance: bytes 16 hh + j < P of rows <= NT - 2
    for hh in (0, 1):
        for r in range(NT - 1):
            cut[32 * hh + r] = 16 * hh + np.arange(16) < P
    acc, A = [], [L.copy() for L in lanes]
    G0 = toeplitz_tile(s, N, 0)
    for p in range(len(lanes)):
        acc.append(as_matrix(A[p]) @ G0)
    for d in range(1, NT):
        G = toeplitz_tile(s, N, d)
        for p in range(len(lanes)):
            j = NT - d
            A[p] = up_enter(A[p], images[p][32 * j: 32 * j + 32])
            if d == NT - 1:
                A[p] = np.where(cut, 0, A[p])
            acc[p] = acc[p] + as_matrix(A[p]) @ G
    return [c.reshape(-1)[:N] for c in acc], NT


def product_split_tile16(a_planes, s, N):
    """The same product on the 16-row tile v_mfma_i32_16x16x64_i8 (K = 64: TWO distances per instruction, two 16-column halves),
    lane for lane as a kernel would run it.  Lane l = (m, kq), m = l & 15 a row of the row group g (output tile kb = 16 g + m), kq = l >> 4:
    kq 0, 1 hold the two 16-byte K halves of the pair's first distance, kq 2, 3 of its second (one chunk row further).  Low walk: pairs
    (0, 1), (2, 3), ...: A = chunk kb - d - (kq >> 1); a pair step moves every 16-lane row by two rows (row_shr:2), group 1 takes its two
    entering rows from group 0's top two.  High walk: pairs (0h, -1), (-2, -3), ...: A = chunk kb + e + (kq >> 1); row_shl:2, group 0
    takes its entering rows from group 1's bottom two.  B lane (n, kq) = bytes sc[32 dist + (16 c + n) - (16 (kq & 1) + j)], the diagonal
    distance split by k' >= i'.  D lane (n, rq): rows 4 rq + j.  Returns (low, high) per plane and the instruction count per plane."""
    NT = tiles(N)
    sN = np.asarray(s, np.int64)
    F = []
    for a in a_planes:
        ap = np.zeros(32 * 36, np.int64); ap[32:32 + N] = a                # chunk index + 1: chunks -1 and NT .. 34 are zero
        F.append(ap.reshape(36, 32))
    lanes_m, lanes_kq = np.arange(64) & 15, np.arange(64) >> 4
    def rows(p, g, sign):                                                   # [64 lanes][16 bytes]
        ch = 16 * g + lanes_m + sign * (lanes_kq >> 1)
        return np.stack([F[p][ch[l] + 1][16 * (lanes_kq[l] & 1): 16 * (lanes_kq[l] & 1) + 16] for l in range(64)])
    def frag(dist0, sign, c, high):                                         # fragment of the pair whose first distance is dist0
        out = np.zeros((64, 16), np.int64)
        for l in range(64):
            n, kq = lanes_m[l], lanes_kq[l]
            dist = dist0 + sign * (kq >> 1)
            kp = 16 * c + n
            for j in range(16):
                ip = 16 * (kq & 1) + j
                v = sN[(32 * dist + kp - ip) % N]
                if dist == 0: v = v if ((kp >= ip) != high) else 0
                out[l, j] = v
        return out
    def mfma16(A, B):
        Am = np.zeros((16, 64), np.int64); Bm = np.zeros((64, 16), np.int64)
        for l in range(64):
            Am[lanes_m[l], 16 * lanes_kq[l]:16 * lanes_kq[l] + 16] = A[l]
            Bm[16 * lanes_kq[l]:16 * lanes_kq[l] + 16, lanes_m[l]] = B[l]
        return Am @ Bm                                                     # [m][n]
    def shift(A0, A1, up):                                                  # one pair step on a plane's two row groups
        N0, N1 = np.zeros_like(A0), np.zeros_like(A1)
        for l in range(64):
            m = lanes_m[l]
            if up:                                                          # row m <- row m - 2; group 1 rows 0, 1 <- group 0 rows 14, 15
                N0[l] = A0[l - 2] if m >= 2 else 0
                N1[l] = A1[l - 2] if m >= 2 else A0[l + 14]
            else:                                                           # row m <- row m + 2; group 0 rows 14, 15 <- group 1 rows 0, 1
                N1[l] = A1[l + 2] if m <= 13 else 0
                N0[l] = A0[l + 2] if m <= 13 else A1[l - 14]
        return N0, N1
    n_pairs, G = (NT + 1) // 2, (NT + 15) // 16
    res, n_instr = [], 0
    for p in range(len(a_planes)):
        acc = np.zeros((2, 2, 2, 16, 16), np.int64)                         # [low / high][group][column half][m][n]
        for high in (0, 1):
            sign = -1 if not high else 1                                    # chunk index moves down in the low walk, up in the high walk
            A = [rows(p, 0, sign), rows(p, 1, sign)]
            for pr in range(n_pairs):
                d0 = 2 * pr
                for g in range(G):
                    if not A[g].any(): continue                             # (the kernel knows from NT and pr which groups still hold rows)
                    for c in range(2):
                        acc[high, g, c] += mfma16(A[g], frag(d0 if not high else -d0, 1 if not high else -1, c, bool(high)))
                        n_instr += 1 if p == 0 else 0
                A[0], A[1] = shift(A[0], A[1], up=not high)
        low = np.zeros(32 * 32, np.int64); hi = np.zeros(32 * 32, np.int64)
        for g in range(2):
            for c in range(2):
                for m in range(16):
                    k0 = 32 * (16 * g + m) + 16 * c
                    low[k0:k0 + 16] = acc[0, g, c, m]; hi[k0:k0 + 16] = acc[1, g, c, m]
        res.append((low[:N], hi[:N]))
    return res, n_instr


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
    tpitch = ((16 * NT + 31) // 32) * 32 + 8
    arr, nat = 16 * tpitch, max((3 * N + 64 + 15) // 16 * 16, 2304)
    print("  LDS per wave: rows in the LDS (until round 4) %d B = two padded chunk matrices + one reversed array; rows in registers %d B "
          "(one array + the natural-order area), k_verify_keys_m %d B (two arrays), on the 16-row tile %d B (+ the planes' byte image)"
          % (2 * 32 * (NT + 64) + arr, nat + arr, nat + 2 * arr, max(nat, 3 * 35 * 32 + 32) + 2 * arr))
    print("  -> matrix clocks %.2fx, lane shifts 1.5x.  Built in round 5 as k_verify_keys_m16 (experiments build): bit-exact and 20 %% SLOWER "
          "as a whole kernel -- more vector instructions per matrix clock, 32-byte store segments (EXPERIMENTS.md round 5 item 9)"
          % (instr16 * 16 / (2 * NT * 32.0)))


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
        if N >= 64:                                         # quotient not wanted: one matrix instruction per distance, one accumulator
            cyc, ncyc = product_cyclic_registers([v & 127, v >> 7], f, N)
            assert np.array_equal((cyc[0] + 128 * cyc[1]) % q, (lin[:N] + lin[N:]) % q), (N, q, "cyclic")
            print("N=%d: modulo x^N - 1 only: %d matrix instructions per plane (%d with the split)  OK" % (N, ncyc, 2 * tiles(N)))
        if N in (821, 167, 33, 509):                        # the 16-row tile, lane for lane (not built: EXPERIMENTS.md round 5)
            p16, n16 = product_split_tile16([v & 127, v >> 7, fp], f, N)
            lo16 = p16[0][0] + 128 * p16[1][0]; hi16 = p16[0][1] + 128 * p16[1][1]
            assert np.array_equal(lo16, lo) and np.array_equal(hi16, hi), (N, q, "16-row tile")
            assert np.array_equal(p16[2][0], low3) and np.array_equal(p16[2][1], high3), (N, "16-row tile, mod p")
            print("N=%d: 16-row tile: %d instructions of 16 clocks per plane (%d of 32 clocks on the 32-row tile)  OK" % (N, n16, 2 * tiles(N)))
        print("N=%d q=%d: %d + %d matrix instructions per item for (f * fq, f * fp); rows in registers: %d fragment reads for the three "
              "planes and no row reads (rows in the LDS: %d fragment + %d row reads); packed-MAC wave instructions of the vector-ALU family: %d  OK"
              % (N, q, n, n3, reads, 2 * reads, 3 * reads, 2 * ((N * N + 127) // 128)))


if __name__ == "__main__":
    main()
