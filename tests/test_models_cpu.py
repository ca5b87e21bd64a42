"""The executable specifications of the kernels' loop structures (tools/*_model.py) are part of the design: DESIGN.md points at them,
so they must keep running.  Each reproduces one kernel family's decomposition in numpy and asserts it against a direct convolution."""
import os
import subprocess
import sys

import pytest

import __graft_entry__ as ge


@pytest.mark.parametrize("model", ["mfma_model.py", "peritem_mfma_model.py", "lane_model.py", "dot8_model.py"])
def test_executable_specification_runs(model):
    out = subprocess.run([sys.executable, os.path.join(ge.ROOT, "tools", model)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "OK" in out.stdout or "ok" in out.stdout.lower(), out.stdout[-500:]


def test_rows_in_registers_model_matches_the_convolution():
    """The per-item kernels' "chunk rows in registers" form (matrix_peritem.hip): lane shifts by one with the half-wave seam cut, the
    low part walking up and the high part walking down, every plane against one fragment per distance."""
    sys.path.insert(0, os.path.join(ge.ROOT, "tools"))
    import numpy as np
    import peritem_mfma_model as pm
    rng = np.random.default_rng(5)
    for N, q in ((64, 32), (95, 2048), (96, 4096), (1024, 8192), (821, 4096)):
        f = rng.integers(-1, 2, N); fq = rng.integers(0, q, N); fp = rng.integers(0, 3, N)
        planes, reads = pm.product_split_registers([fq & 127, fq >> 7, fp], f, N)
        assert reads == 2 * pm.tiles(N) - 1
        lin = np.convolve(f, fq); lin = np.concatenate([lin, np.zeros(2 * N - len(lin), np.int64)])
        lo = planes[0][0] + 128 * planes[1][0]; hi = planes[0][1] + 128 * planes[1][1]
        assert np.array_equal(lo, lin[:N]) and np.array_equal(hi, lin[N:]), (N, q)
        lin3 = np.convolve(f, fp); lin3 = np.concatenate([lin3, np.zeros(2 * N - len(lin3), np.int64)])
        assert np.array_equal(planes[2][0], lin3[:N]) and np.array_equal(planes[2][1], lin3[N:]), N
        # product 3 of verifyKeysInputs as p (fq * g): the same planes against g's fragments, the factor in the epilogue
        g = rng.integers(-1, 2, N)
        pg, _ = pm.product_split_registers([fq & 127, fq >> 7], g, N)
        lo3 = 3 * (pg[0][0] + 128 * pg[1][0]); hi3 = 3 * (pg[0][1] + 128 * pg[1][1])
        ref = np.convolve(g, (3 * fq) % q); ref = np.concatenate([ref, np.zeros(2 * N - len(ref), np.int64)])
        assert np.array_equal((lo3 + hi3) % q, (ref[:N] + ref[N:]) % q) and np.array_equal((-hi3) % q, (-ref[N:]) % q), (N, q)


def test_cyclic_rows_model_matches_the_convolution_modulo_x_n_minus_1():
    """`pi_product_cyc` (Newton rounds, public key): one matrix instruction per tile distance, the wrapped terms through rows of the
    copy moved up by 32 NT - N places that enter at row 0; every N class (multiple of 32, one short of it, one past it, two tiles)."""
    sys.path.insert(0, os.path.join(ge.ROOT, "tools"))
    import numpy as np
    import peritem_mfma_model as pm
    rng = np.random.default_rng(6)
    for N in (64, 65, 95, 96, 97, 127, 167, 509, 677, 701, 821, 992, 993, 1023, 1024):
        a = rng.integers(0, 128, N); b = rng.integers(-64, 64, N); s = rng.integers(-1, 2, N)
        (c, c2), n = pm.product_cyclic_registers([a, b], s, N)
        assert n == pm.tiles(N)
        for x, got in ((a, c), (b, c2)):
            lin = np.convolve(x, s); lin = np.concatenate([lin, np.zeros(2 * N - len(lin), np.int64)])
            assert np.array_equal(got, lin[:N] + lin[N:]), N
