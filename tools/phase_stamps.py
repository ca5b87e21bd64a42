#!/usr/bin/env python3
"""Phase timeline of the matrix-core kernels from a -DNTRU_STAMPS build (diagnostic, never shipped):
     make -C ntru-circom_amd/csrc EXTRA=-DNTRU_STAMPS OBJDIR=../lib/ab/obj_stamps OUT=../lib/ab/libntru_stamps.so
     NTRU_ENGINE_LIB=$PWD/ntru-circom_amd/lib/ab/libntru_stamps.so python tools/phase_stamps.py
Prints, per kernel, the median duration (cycles of s_memtime = 100 MHz ticks x ... shader clock) of each phase over
workgroups and waves, for row-block iterations 2..4 of every workgroup."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
eng = pkg.Engine(0)
lib = C.CDLL(os.environ["NTRU_ENGINE_LIB"])
dev = torch.device("cuda:0")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_kernel_path(4)      # two free-running workgroups per CU (the stamp buffer is laid out for four waves per workgroup)
N, q, B = 821, 4096, 1 << 20
h = torch.randint(0, q, (N,), dtype=torch.int32, device=dev).to(torch.int16)
f = torch.randint(-1, 2, (N,), dtype=torch.int8, device=dev); fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev)
r = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev); m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev)
e = torch.empty((B, N), dtype=torch.int16, device=dev); qe = torch.empty_like(e)
v = torch.empty((B, N), dtype=torch.uint8, device=dev); q2 = torch.empty_like(v); q1 = torch.empty_like(e); r1 = torch.empty_like(e)
SLOTS, BLK = 24, 6
buf = np.zeros((1024, 8, BLK, SLOTS), np.uint64)
def read(which):       # every stamping translation unit has its own buffer: "enc" (matrix_encrypt.hip) / "dec" (matrix_decrypt.hip)
    torch.cuda.synchronize()
    assert getattr(lib, "ntru_debug_read_stamps_" + which)(buf.ctypes.data_as(C.c_void_p)) == 0
    return buf.copy()
def report(name, st, labels, nblocks, waves=slice(0, 4)):
    st = st[:nblocks, waves].astype(np.int64)
    print("==", name, "(median over workgroups x waves, iterations 2..4; cycles)")
    for it in (2, 3, 4):
        row = []
        for a, b, lab in labels:
            d = st[:, :, it, b] - st[:, :, it, a]
            d = d[(st[:, :, it, a] > 0) & (st[:, :, it, b] > 0)]
            row.append("%s %d" % (lab, int(np.median(d)) if d.size else -1))
        tot = st[:, :, it + 1, 0] - st[:, :, it, 0]
        print(" iter", it, "|", " | ".join(row), "| whole row block", int(np.median(tot[tot > 0])))
    # phase offset between the two workgroups of a CU (b and b + nblocks/2) at iteration 3
    half = nblocks // 2
    off = (st[half:nblocks, 0, 3, 0] - st[:half, 0, 3, 0])
    per = np.median(st[:, 0, 4, 0] - st[:, 0, 3, 0])
    print(" start-of-row-block offset between workgroups b and b+%d: median %d cycles (period %d)" % (half, int(np.median(off)), int(per)))
for _ in range(2):
    eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), qe.data_ptr())
st = read("enc")
report("k_encrypt_m", st, [(0, 1, "wait barrier1"), (1, 16, "stage r"), (16, 2, "stage m"), (2, 3, "wait barrier2"), (3, 4, "loops s1"), (4, 5, "epilogue s1"),
                           (5, 6, "loops s2"), (6, 7, "epilogue s2")], 512)
eng.set_kernel_path(5)      # k_encrypt_md: operands by direct-to-LDS loads (the default encrypt kernel)
for _ in range(2):
    eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), qe.data_ptr())
st = read("enc")
report("k_encrypt_md", st, [(0, 1, "wait barrier1"), (1, 2, "request m, r in place"), (2, 3, "wait barrier2"), (3, 4, "loops s1"), (4, 5, "(m wait, barrier) epilogue s1"),
                            (5, 6, "loops s2"), (6, 7, "(barrier, request r) epilogue s2")], 512)
eng.set_kernel_path(4)
for _ in range(2):
    eng.decrypt_batch_dev(N, q, 3, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr(), q1.data_ptr(), r1.data_ptr(), q2.data_ptr())
st = read("dec")
report("k_decrypt_m", st, [(0, 1, "wait b1"), (1, 2, "stage"), (2, 3, "wait b2"), (3, 4, "P1 loops s1"), (4, 5, "P1 epi s1"), (5, 6, "P1 loops s2"),
                           (6, 7, "P1 epi s2"), (7, 8, "wait b3"), (8, 9, "expand"), (9, 10, "wait b4"), (10, 11, "P2 loops s1"), (11, 12, "P2 epi s1"),
                           (12, 13, "P2 loops s2"), (13, 14, "P2 epi s2")], 512)

# the lock-step kernel: one workgroup of two four-wave groups per CU; per group (group 1 runs one phase behind)
eng.set_kernel_path(5)
for _ in range(2):
    eng.decrypt_batch_dev(N, q, 3, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr(), q1.data_ptr(), r1.data_ptr(), q2.data_ptr())
st = read("dec")
lab8 = [(0, 1, "wait b1"), (1, 2, "stage"), (2, 3, "wait b2"), (3, 4, "P1 loops s1"), (4, 5, "P1 (phase barrier) epi s1"), (5, 6, "(barrier) P1 loops s2"),
        (6, 7, "P1 (barrier) epi s2"), (7, 8, "wait b3"), (8, 9, "expand"), (9, 10, "wait b4"), (10, 11, "P2 loops s1"), (11, 12, "P2 (barrier) epi s1"),
        (12, 13, "(barrier) P2 loops s2"), (13, 14, "P2 (barrier) epi s2")]
report("k_decrypt_m8 group 0", st, lab8, 256, slice(0, 4))
report("k_decrypt_m8 group 1", st, lab8, 256, slice(4, 8))
eng.set_kernel_path(8)      # needs EXTRA="-DNTRU_STAMPS -DNTRU_EXPERIMENTS"
for _ in range(2):
    eng.decrypt_batch_dev(N, q, 3, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr(), q1.data_ptr(), r1.data_ptr(), q2.data_ptr())
st = read("dec")
report("k_decrypt_m8d group 0 (rows by direct-to-LDS loads)", st, lab8, 256, slice(0, 4))
report("k_decrypt_m8d group 1", st, lab8, 256, slice(4, 8))
