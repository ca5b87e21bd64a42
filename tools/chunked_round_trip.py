#!/usr/bin/env python3
"""Does a round trip get cheaper when decryptBits follows encryptBits chunk by chunk (the ciphertexts still in the 256 MB memory-side
cache) instead of batch by batch?  Same kernels, same 2^20 items, all witness arrays; chunk = 2^k items per encrypt / decrypt pair."""
import json
import sys
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

pkg = ge.load_package()
dev = torch.device("cuda", 0)
o, h_np, f_np, fp_np = bench.load_key("n821_q4096")
N, q, p, d = o["N"], o["q"], o["p"], o["dr"]
B = 1 << 20
r, m = bench.make_inputs(torch, dev, B, N, d, 20240)
h = torch.from_numpy(h_np.view(np.int16)).to(dev); f = torch.from_numpy(f_np).to(dev); fp = torch.from_numpy(fp_np).to(dev)
b16 = lambda: torch.empty((B, N), dtype=torch.int16, device=dev)
b8 = lambda: torch.empty((B, N), dtype=torch.uint8, device=dev)
e, qe, v, q1, r1, q2 = b16(), b16(), b8(), b16(), b16(), b8()
eng = pkg.Engine(0)
stream = torch.cuda.current_stream()
eng.set_stream(stream.cuda_stream)


def step(logc):
    C = 1 << logc
    for o_ in range(0, B, C):
        o8, o16 = o_ * N, o_ * N * 2
        eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr() + o8, m.data_ptr() + o8, C, e.data_ptr() + o16, qe.data_ptr() + o16)
        eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr() + o16, C, v.data_ptr() + o8, q1.data_ptr() + o16,
                              r1.data_ptr() + o16, q2.data_ptr() + o8)


ref = None
for logc in (20, 18, 17, 16, 15, 20):
    for _ in range(3):
        step(logc)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(10):
        step(logc)
    ev1.record(stream)
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 10
    chk = int(v.to(torch.int64).sum().item()) ^ int(q1.to(torch.int64).sum().item())
    ref = chk if ref is None else ref
    print(json.dumps({"chunk_log2": logc, "ms_per_2^20_round_trips": ms, "round_trips_per_s_M": B / ms / 1e3, "same_results": chk == ref}))
