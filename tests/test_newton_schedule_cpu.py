"""The Newton lifting of ntru_invert_key_batch_dev, restated in plain numpy (CPU only): precision schedule halved backwards from log2 q,
every round in its lifted form  e = (f v - 1) / 2^kb,  v <- v - 2^kb (e v mod 2^(m - kb))  -- against the oracle's restatement of the
reference's polyInv (index.js:491-514: log2(q) - 1 rounds of v <- 2 v - f v^2).  The inverse modulo q is unique, so the two must agree
coefficient by coefficient; the device kernels (k_newton_round_m, or k_product_tern_m + k_polymul_m) compute exactly these rounds."""
import numpy as np
import pytest

from oracle import ntru_keygen as kg


def schedule(k):
    """bits of v before every round and after the last one: k halved backwards down to 1 (csrc/keygen_sampler_pack.hip: bits_of)."""
    rounds = 0
    while (1 << rounds) < k:
        rounds += 1
    bits = [0] * (rounds + 1)
    bits[rounds] = k
    for r in range(rounds, 0, -1):
        bits[r - 1] = (bits[r] + 1) // 2
    return bits


def cyc(a, b, N, mod):
    """a * b in Z_mod[x] / (x^N - 1)."""
    full = np.convolve(np.asarray(a, dtype=object), np.asarray(b, dtype=object))
    out = np.zeros(N, dtype=object)
    for i, c in enumerate(full):
        out[i % N] += c
    return np.array([int(c) % mod for c in out], dtype=np.int64)


def lifted_inverse(f, N, q):
    k = q.bit_length() - 1
    v = np.zeros(N, np.int64)
    inv2 = kg.poly_inv(f, N, 2)
    v[:len(inv2)] = np.asarray(inv2) % 2
    bits = schedule(k)
    assert bits[0] == 1 and bits[-1] == k
    for kb, m in zip(bits[:-1], bits[1:]):
        assert kb < m <= 2 * kb and (k > 14 or kb <= 7)      # a round at most doubles; v stays below 128 while log2 q <= 14
        mr, me = 1 << m, 1 << (m - kb)
        assert int(v.max()) < (1 << kb)
        fv = cyc(f, v, N, mr)
        assert fv[0] % (1 << kb) == 1 and not (fv[1:] % (1 << kb)).any()      # f v = 1 modulo 2^kb
        e = fv.copy(); e[0] = (e[0] - 1) % mr
        e = (e >> kb) % me
        w = cyc(e, v % me, N, me)
        v = (v - (w << kb)) % mr
    return v


def test_schedule_shapes():
    assert schedule(12) == [1, 2, 3, 6, 12] and schedule(13) == [1, 2, 4, 7, 13] and schedule(11) == [1, 2, 3, 6, 11]
    assert schedule(16) == [1, 2, 4, 8, 16] and schedule(1) == [1] and schedule(2) == [1, 2] and schedule(3) == [1, 2, 3]
    for k in range(1, 17):
        s = schedule(k)
        assert len(s) - 1 == (0 if k == 1 else (k - 1).bit_length())      # as many rounds as doubling from 1 takes


@pytest.mark.parametrize("N,q", [(11, 32), (17, 2048), (23, 4096), (31, 8192), (37, 65536), (13, 4), (29, 128)])
def test_lifted_rounds_equal_the_reference_inverse(N, q):
    rng = np.random.default_rng(N * q)
    done = 0
    for _ in range(60):
        f = rng.integers(-1, 2, N)
        if not kg.is_unit(f, N, 2):
            continue
        want = np.zeros(N, np.int64)
        ref = np.asarray(kg.poly_inv(f, N, q)) % q
        want[:len(ref)] = ref
        got = lifted_inverse(f, N, q)
        assert np.array_equal(got, want), (N, q, f.tolist())
        one = np.zeros(N, np.int64); one[0] = 1
        assert np.array_equal(cyc(f, got, N, q), one)
        done += 1
        if done == 6:
            break
    assert done >= 3
