"""The degree bounds k_invert_key skips words by (csrc/keygen_sampler_pack.hip), checked on a plain-Python run of the same division
steps (CPU only).  Before step n of the 2N - 1 Bernstein-Yang steps on ff = 1 - x^N, gg = rev_{N-1}(f), vv = 0, ww = 1, delta = 1:
    2 deg f <= 2N - 1 - n + delta      2 deg g <= 2N - 1 - n - delta      2 deg v <= n - 1 + delta      2 deg w <= n + 1 - delta
and the kernel's word tests derive from them with the wave maximum of |delta| taken every 16th step: f and g are zero above
(2N - 1 - n0 + mx) / 2 for every n >= n0 (n - |delta| never decreases), v and w -- as read AND as written by steps n0 .. n0 + 15 -- above
(n0 + mx + 34) / 2.  Also: the inverse comes out of vv, reversed and scaled by f(0), exactly as the kernel extracts it."""
import numpy as np
import pytest

from oracle import ntru_keygen as kg


def deg(a):
    nz = np.nonzero(a)[0]
    return int(nz[-1]) if len(nz) else -10 ** 6                # the zero polynomial: below every bound


def run(fpoly, N, P, check):
    """The kernel's step, coefficient arrays of length 2N + 2 modulo P; check(n, f, g, v, w, delta) before every step and after the last."""
    L = 2 * N + 2
    f = np.zeros(L, np.int64); f[0] = 1; f[N] = P - 1
    g = np.zeros(L, np.int64); g[:N] = (np.asarray(fpoly)[::-1]) % P
    v = np.zeros(L, np.int64); w = np.zeros(L, np.int64); w[0] = 1
    delta = 1
    for n in range(2 * N - 1):
        check(n, f, g, v, w, delta)
        fc, gc = int(f[0]), int(g[0])
        swap = delta > 0 and gc != 0
        delta = (-delta if swap else delta) + 1
        xv = np.roll(v, 1); xv[0] = 0                         # v = x v
        if swap:
            f, g = g.copy(), f.copy()
            v, w = w.copy(), xv
        else:
            v = xv
        # the new g(0) cancels against a multiple of the new f; g and w are scaled by the unit 1 / f(0) (GF(2): 1)
        inv_f0 = pow(int(f[0]), -1, P) if f[0] else 0
        c = (P - (int(g[0]) * inv_f0) % P) % P
        g = (g + c * f) % P
        w = (w + c * v) % P
        assert g[0] == 0
        g = np.roll(g, -1); g[-1] = 0                          # g = g / x
    check(2 * N - 1, f, g, v, w, delta)
    return f, v, delta


@pytest.mark.parametrize("N,P", [(11, 2), (11, 3), (23, 2), (23, 3), (37, 3), (64, 2), (67, 3)])
def test_degree_bounds_hold_at_every_step_and_the_inverse_comes_out(N, P):
    rng = np.random.default_rng(1000 * N + P)
    units = 0
    for trial in range(14):
        fpoly = rng.integers(-1, 2, N)
        if trial == 0:
            fpoly = np.zeros(N, np.int64)                     # g = 0 from the start: delta grows for ever, f keeps degree N
        if trial == 1:
            fpoly = np.zeros(N, np.int64); fpoly[0] = 1; fpoly[1] = P - 1      # 1 - x: divides x^N - 1
        seen = {}

        def check(n, f, g, v, w, delta):
            assert 2 * deg(f) <= 2 * N - 1 - n + delta and 2 * deg(g) <= 2 * N - 1 - n - delta, (n, delta)
            assert 2 * deg(v) <= n - 1 + delta and 2 * deg(w) <= n + 1 - delta, (n, delta)
            seen[n] = (abs(delta), max(deg(f), deg(g)), max(deg(v), deg(w)))

        f, v, delta = run(fpoly, N, P, check)
        # the kernel's sampled bounds: |delta| of step n0 = 16 j serves steps n0 .. n0 + 15, reads and writes
        for n0 in range(0, 2 * N - 1, 16):
            mx = seen[n0][0]
            for n in range(n0, min(n0 + 16, 2 * N - 1)):
                assert seen[n][1] <= (2 * N - 1 - n0 + mx) >> 1
                assert max(seen[n][2], seen[n + 1][2]) <= (n0 + mx + 34) >> 1
        unit = deg(f) == 0 and f[0] != 0
        assert unit == kg.is_unit(fpoly, N, P)
        assert max(deg(f), 0) <= abs(delta) >> 1               # the final bound the kernel checks `rest` under (n = 2N - 1)
        if unit:
            inv = (int(f[0]) * v[:N][::-1]) % P if P == 3 else v[:N][::-1] % 2          # inverse[i] = f(0)^-1 vv[N-1-i]; in GF(3) f(0)^-1 = f(0)
            want = np.zeros(N, np.int64)
            ref = np.asarray(kg.poly_inv(fpoly, N, P)) % P
            want[:len(ref)] = ref
            assert np.array_equal(inv, want), (N, P, fpoly.tolist())
            units += 1
    assert units >= 2


def test_gf3_multiply_add_on_nonzero_and_sign_planes():
    """k_invert_key<3>'s r = g + cm f on (non-zero, sign) bit planes, bit for bit as the kernel computes it (madd3), for every
    g, f, cm in GF(3) and every value of the sign bits that lie under a clear non-zero bit (garbage by design)."""
    import itertools
    for g, f, cm in itertools.product(range(3), repeat=3):
        for junk_g, junk_f in itertools.product((0, 1), repeat=2):
            g0, g1 = int(g != 0), (int(g == 2) if g else junk_g)
            f0, f1 = int(f != 0), (int(f == 2) if f else junk_f)
            mnz, m2 = int(cm != 0), int(cm == 2)
            tn, ts = f0 & mnz, f1 ^ m2
            x = g1 ^ ts
            r0 = (g0 ^ tn) | (g0 & tn & (1 - x))
            p = g1 ^ tn
            r1 = (g0 & p) | ((1 - g0) & ts)
            assert ((1 + r1) if r0 else 0) == (g + cm * f) % 3, (g, f, cm, junk_g, junk_f)
