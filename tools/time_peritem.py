#!/usr/bin/env python3
"""Launch times of the per-item kernels on synthetic operands (no result check: for same-device A/B of library builds, also of
timing-only experiments):   NTRU_ENGINE_LIB=... python tools/time_peritem.py [logB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
eng = pkg.Engine(0)
dev = torch.device("cuda:0")
eng.set_stream(torch.cuda.current_stream().cuda_stream)
N, q, B = 821, 4096, 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
kf = torch.randint(-1, 2, (B, N), dtype=torch.int8, device=dev); kg = torch.randint(-1, 2, (B, N), dtype=torch.int8, device=dev)
kfq = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev).to(torch.int16); kfp = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev)
kh = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev).to(torch.int16)
o16 = [torch.empty((B, N), dtype=torch.int16, device=dev) for _ in range(4)]; o8 = [torch.empty((B, N), dtype=torch.uint8, device=dev) for _ in range(2)]
fl = torch.empty(B, dtype=torch.uint8, device=dev)
def timed(fn, reps=20):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
v = lambda: eng.verify_keys_batch_dev(N, q, 3, kf.data_ptr(), kg.data_ptr(), kfq.data_ptr(), kfp.data_ptr(), kh.data_ptr(), B, o16[0].data_ptr(),
                                      o16[1].data_ptr(), o8[0].data_ptr(), o8[1].data_ptr(), o16[2].data_ptr(), o16[3].data_ptr(), fl.data_ptr())
p = lambda: eng.polymul_split_dev(N, q, kfq.data_ptr(), kh.data_ptr(), B, o16[0].data_ptr(), o16[1].data_ptr())
print("lib %s  2^%d items: verify_keys %.4f ms, polymul %.4f ms" % (os.path.basename(os.environ.get("NTRU_ENGINE_LIB", "default")), B.bit_length() - 1, timed(v), timed(p)))
