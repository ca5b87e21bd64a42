"""Lane-level numpy model of the wave-per-item polymul-split kernel (design check, not product code).

Mirrors csrc/ntru_kernels.hip step for step: every "lane" owns 2K consecutive outputs as K packed u16
pairs, the windowed operand lives in an interleaved array EO[u] = (E[u], O[u]) with
E[u] = (bc[2u], bc[2u+1]), O[u] = (bc[2u-1], bc[2u]), bc = b extended cyclically with period N, and the
low half c[k] of the linear product is recovered as snapshot-before-own-block + in-block triangle.
"""
import numpy as np


def polymul_split_model(a, b, N, K, mod_mask=None, p=None):
    a = np.asarray(a, dtype=np.int64); b = np.asarray(b, dtype=np.int64)
    nl = -(-N // (2 * K))
    assert N >= 2 * K
    OFF = (N + 1) // 2
    M16 = 0xFFFF
    bc = lambda j: int(b[j % N])
    nU = K * nl + OFF
    E = np.zeros((nU, 2), np.int64); O = np.zeros((nU, 2), np.int64)
    for x in range(nU):
        u = x - OFF
        E[x] = (bc(2 * u), bc(2 * u + 1)); O[x] = (bc(2 * u - 1), bc(2 * u))
    apad = np.zeros(2 * K * nl, np.int64); apad[:N] = a
    T = np.zeros((nl, K, 2), np.int64); snap = np.zeros((nl, K, 2), np.int64)
    WE = np.zeros((nl, K, 2), np.int64); WO = np.zeros((nl, K, 2), np.int64)
    lanes = np.arange(nl)
    # prologue: window for s = 0: phys[(x) mod K] holds EO[K*l + x], x in [0, K)
    for x in range(K):
        WE[:, x % K] = E[K * lanes + x + OFF]; WO[:, x % K] = O[K * lanes + x + OFF]
    for m in range(nl):
        snap[m] = T[m]                                  # lane m snapshots before its own block
        base = K * lanes - K * m                       # index of x = -s at sigma = 0
        nE = [E[base - 1 - sg + OFF] for sg in range(K)]
        nO = [O[base - 1 - sg + OFF] for sg in range(K)]
        for sg in range(K):
            i = 2 * (K * m + sg)
            alo, ahi = apad[i], apad[i + 1]
            for t in range(K):
                T[:, t] = (T[:, t] + alo * WE[:, (t - sg) % K]) & M16
            for t in range(K):
                T[:, t] = (T[:, t] + ahi * WO[:, (t - sg) % K]) & M16
            WE[:, (-sg - 1) % K] = nE[sg]; WO[:, (-sg - 1) % K] = nO[sg]
    # in-block triangle: diag[d] = sum_{u<=d} a[k0+u] * b[d-u]
    ZE = [E[x + OFF].copy() for x in range(K)]; ZO = [O[x + OFF].copy() for x in range(K)]
    ZO[0][0] = 0
    diag = np.zeros((nl, K, 2), np.int64)
    for sg in range(K):
        alo = apad[2 * K * lanes + 2 * sg][:, None]; ahi = apad[2 * K * lanes + 2 * sg + 1][:, None]
        for t in range(sg, K):
            diag[:, t] = (diag[:, t] + alo * ZE[t - sg]) & M16
            diag[:, t] = (diag[:, t] + ahi * ZO[t - sg]) & M16
    low = (snap + diag) & M16
    high = (T - low) & M16
    Tf = T.reshape(-1)[:N]; hf = high.reshape(-1)[:N]
    if p is None:
        return (-hf) & mod_mask, Tf & mod_mask
    return (p - hf % p) % p, Tf % p


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ".")
    from oracle import ntru_oracle as orc
    rng = np.random.default_rng(1)
    for N, K, q in [(17, 1, 32), (17, 3, 32), (167, 3, 128), (509, 5, 2048), (821, 7, 4096), (701, 7, 8192), (20, 3, 64)]:
        a = rng.integers(0, q, N); b = rng.integers(0, q, N)
        qo, ro = orc.polymul_split_batch(N, q, a, b)
        qm, rm = polymul_split_model(a, b, N, K, mod_mask=q - 1)
        assert (qo[0] == qm).all() and (ro[0] == rm).all(), (N, K, q)
        a3 = rng.integers(0, 3, N); b3 = rng.integers(0, 3, N)
        qo, ro = orc.polymul_split_batch(N, 3, a3, b3)
        qm, rm = polymul_split_model(a3, b3, N, K, p=3)
        assert (qo[0] == qm).all() and (ro[0] == rm).all(), (N, K, 3)
        print("ok", N, K, q)
