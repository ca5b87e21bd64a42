#!/usr/bin/env python3
"""Phase timeline of the row-image kernels (matrix_rowimage.hip) from a -DNTRU_STAMPS build (diagnostic, never shipped):
     make -C ntru-circom_amd/csrc EXTRA=-DNTRU_STAMPS OBJDIR=../lib/ab/obj_stamps OUT=../lib/ab/libntru_stamps.so
     NTRU_ENGINE_LIB=$PWD/ntru-circom_amd/lib/ab/libntru_stamps.so python tools/phase_stamps_rowimage.py [encrypt|decrypt]
Median over workgroups of the s_memtime difference between phase boundaries, per WAVE (the eight waves of the workgroup own
different strips), row-block iterations 2..4."""
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
eng.set_kernel_path(10)
which = sys.argv[1] if len(sys.argv) > 1 else "encrypt"
N, q, B = 821, 4096, 1 << 20
h = torch.randint(0, q, (N,), dtype=torch.int32, device=dev).to(torch.int16)
f = torch.randint(-1, 2, (N,), dtype=torch.int8, device=dev); fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev)
r = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev); m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev)
e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev).to(torch.int16); qe = torch.empty_like(e)
v = torch.empty((B, N), dtype=torch.uint8, device=dev); q2 = torch.empty_like(v); q1 = torch.empty_like(e); r1 = torch.empty_like(e)
SLOTS, BLK = 24, 6
buf = np.zeros((1024, 8, BLK, SLOTS), np.uint64)
def read():
    torch.cuda.synchronize()
    assert lib.ntru_debug_read_stamps_rowimage(buf.ctypes.data_as(C.c_void_p)) == 0
    return buf.copy()
def report(name, st, labels, nblocks):
    st = st[:nblocks].astype(np.int64)
    print("==", name, "(median over workgroups, iteration 3; cycles)")
    for w in range(8):
        row = []
        for a, b, lab in labels:
            d = st[:, w, 3, b] - st[:, w, 3, a]
            d = d[(st[:, w, 3, a] > 0) & (st[:, w, 3, b] > 0)]
            row.append("%s %d" % (lab, int(np.median(d)) if d.size else -1))
        tot = st[:, w, 4, 0] - st[:, w, 3, 0]
        print(" wave", w, "|", " | ".join(row), "| whole row block", int(np.median(tot[tot > 0])))
if which == "encrypt":
    for _ in range(2):
        eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), qe.data_ptr())
    print(eng.last_kernel())
    report("k_encrypt_w", read(), [(0, 1, "barrier B"), (1, 2, "loops to diagonal"), (2, 3, "diagonal block + drain"), (3, 4, "loops after"),
                                    (4, 5, "request m"), (5, 6, "barrier A"), (6, 7, "images"), (7, 8, "stage next r")], 256)
else:
    for _ in range(2):
        eng.decrypt_batch_dev(N, q, 3, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr(), q1.data_ptr(), r1.data_ptr(), q2.data_ptr())
    print(eng.last_kernel())
    report("k_decrypt_w", read(), [(0, 1, "barrier B"), (1, 4, "P1 loops (+drain)"), (4, 6, "barrier A"), (6, 7, "images 1 + lift"), (7, 9, "barrier C"),
                                    (9, 12, "P2 loops (+drain)"), (12, 14, "barrier D"), (14, 15, "images 2"), (15, 16, "stage next e")], 256)
