import sys, time, numpy as np, ctypes as C
sys.path.insert(0, '.')
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
eng = pkg.Engine(0)
o, h, f, fp = bench.load_key("n821_q4096")
N, q, p, d = o["N"], o["q"], o["p"], o["dr"]
B = 1 << 17
lib = eng._lib
def pinned(n, dt):
    ptr = lib.ntru_host_alloc(n * np.dtype(dt).itemsize)
    return np.ctypeslib.as_array((C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr)).view(dt)
m = pinned(B * N, np.uint8); m[:] = np.random.default_rng(1).integers(0, 2, B * N, dtype=np.uint8)
value = pinned(B * N, np.uint8)
key = np.arange(8, dtype=np.uint32)
P = lambda a: a.ctypes.data_as(C.c_void_p)
for it in range(3):
    t0 = time.perf_counter()
    rc = lib.ntru_pipeline_batch(eng._h, N, q, p, P(h), P(f), P(fp), P(key), 0, d, d, None, P(m), B, None, None, P(value), None)
    print("rc", rc, "ms", (time.perf_counter() - t0) * 1e3)
