#!/usr/bin/env python3
"""Phase timeline of one item in k_verify_keys_m and k_polymul_m from a -DNTRU_STAMPS build (diagnostic, never shipped):
     make -C ntru-circom_amd/csrc EXTRA=-DNTRU_STAMPS OBJDIR=../lib/ab/obj_stamps OUT=../lib/ab/libntru_stamps.so
     NTRU_ENGINE_LIB=$PWD/ntru-circom_amd/lib/ab/libntru_stamps.so python tools/phase_stamps_peritem.py
Median duration of each phase (shader clocks of s_memtime) over workgroups x waves for the 3rd-5th item of every wave."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
eng = pkg.Engine(0)
lib = C.CDLL(os.environ["NTRU_ENGINE_LIB"])
dev = torch.device("cuda:0")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
N, B = 821, 1 << 16
buf = np.zeros((1024, 8, 6, 24), np.uint64)
labels = [(0, 1, "operands arrive + shift"), (1, 2, "digit planes"), (2, 3, "reversed arrays"), (3, 4, "matrix loops"),
          (4, 5, "result stores issued + fence")]
def show(name, st, labels, last):
    print("==", name)
    for it in (2, 3, 4):
        row = []
        for x, y, lab in labels:
            d = st[:, :, it, y] - st[:, :, it, x]
            d = d[(st[:, :, it, x] > 0) & (st[:, :, it, y] > 0)]
            row.append("%s %d" % (lab, int(np.median(d)) if d.size else -1))
        tot = st[:, :, it + 1, 0] - st[:, :, it, 0]
        print(" item", it, "|", " | ".join(row), "| whole item", int(np.median(tot[tot > 0])))

# k_verify_keys_m on synthetic per-item keys (values do not matter for the timeline)
Bk, q = 1 << 16, 4096
kf = torch.randint(-1, 2, (Bk, N), dtype=torch.int8, device=dev); kg = torch.randint(-1, 2, (Bk, N), dtype=torch.int8, device=dev)
kfq = torch.randint(0, q, (Bk, N), dtype=torch.int32, device=dev).to(torch.int16); kfp = torch.randint(0, 3, (Bk, N), dtype=torch.uint8, device=dev)
kh = torch.randint(0, q, (Bk, N), dtype=torch.int32, device=dev).to(torch.int16)
o16 = [torch.empty((Bk, N), dtype=torch.int16, device=dev) for _ in range(4)]; o8 = [torch.empty((Bk, N), dtype=torch.uint8, device=dev) for _ in range(2)]
fl = torch.empty(Bk, dtype=torch.uint8, device=dev)
for _ in range(2):
    eng.verify_keys_batch_dev(N, q, 3, kf.data_ptr(), kg.data_ptr(), kfq.data_ptr(), kfp.data_ptr(), kh.data_ptr(), Bk, o16[0].data_ptr(), o16[1].data_ptr(),
                              o8[0].data_ptr(), o8[1].data_ptr(), o16[2].data_ptr(), o16[3].data_ptr(), fl.data_ptr())
torch.cuda.synchronize()
assert lib.ntru_debug_read_stamps_pi(buf.ctypes.data_as(C.c_void_p)) == 0
show("k_verify_keys_m, N = %d, q = %d (%s)" % (N, q, eng.last_kernel()), buf[:768, :2].astype(np.int64),
     [(0, 1, "arrays of f and g"), (1, 2, "planes fq lo/hi, fp"), (2, 3, "the loop (5 plane products)"), (3, 4, "P1 epilogue"), (4, 5, "P2 epilogue"),
      (5, 6, "P3 stores"), (6, 7, "h comparison")], 7)

for mod in (16, 4096):
    a = torch.randint(0, mod, (B, N), dtype=torch.int32, device=dev).to(torch.int16)
    b = torch.randint(0, mod, (B, N), dtype=torch.int32, device=dev).to(torch.int16)
    quot = torch.empty_like(a); rem = torch.empty_like(a)
    for _ in range(2):
        eng.polymul_split_dev(N, mod, a.data_ptr(), b.data_ptr(), B, quot.data_ptr(), rem.data_ptr())
    torch.cuda.synchronize()
    assert lib.ntru_debug_read_stamps_pi(buf.ctypes.data_as(C.c_void_p)) == 0
    st = buf[:768, :2].astype(np.int64)          # 6 workgroups per CU x 256 CUs > 1024: the first 768 workgroups, both waves
    print("== k_polymul_m, N = %d, modulus %d (%s), kernel %s" % (N, mod, "one digit plane" if mod <= 256 else "two digit planes: three plane products", eng.last_kernel()))
    for it in (2, 3, 4):
        row = []
        for x, y, lab in labels:
            d = st[:, :, it, y] - st[:, :, it, x]
            d = d[(st[:, :, it, x] > 0) & (st[:, :, it, y] > 0)]
            row.append("%s %d" % (lab, int(np.median(d)) if d.size else -1))
        tot = st[:, :, it + 1, 0] - st[:, :, it, 0]
        print(" item", it, "|", " | ".join(row), "| whole item", int(np.median(tot[tot > 0])))

# k_newton_round_m: the last round of the N = 821, q = 4096 schedule (6 -> 12 bits) on synthetic rows (values do not matter)
f8 = torch.randint(-1, 2, (B, N), dtype=torch.int8, device=dev)
v16 = torch.randint(0, 64, (B, N), dtype=torch.int32, device=dev).to(torch.int16)
lib.ntru_debug_newton_round.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
for _ in range(2):
    assert lib.ntru_debug_newton_round(eng._h, N, 6, 12, f8.data_ptr(), v16.data_ptr(), B) == 0
torch.cuda.synchronize()
assert lib.ntru_debug_read_stamps_pi(buf.ctypes.data_as(C.c_void_p)) == 0
show("k_newton_round_m, N = %d, 6 -> 12 bits (four waves per SIMD)" % N, buf[:1024, :2].astype(np.int64),
     [(0, 1, "operands + digits"), (1, 2, "array of f + image of v"), (2, 3, "f v (26 distances)"), (3, 4, "e, array of v, image of e"),
      (4, 5, "e v (26 distances)"), (5, 6, "lift + stores")], 6)
