#!/usr/bin/env python3
"""Do host-to-device and device-to-host copies overlap on this box?  Pinned buffers, two streams: each direction alone, then both
at once.  (The host pipeline of ntru_host.hip puts the upload of chunk k+1 and the download of chunk k-1 on two streams.)"""
import json
import time

import torch

dev = torch.device("cuda", 0)
n = 256 << 20
h_up = torch.empty(n, dtype=torch.uint8).pin_memory(); h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_up = torch.empty(n, dtype=torch.uint8, device=dev); d_dn = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, dn, reps=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1): d_up.copy_(h_up, non_blocking=True)
        if dn:
            with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


run(True, True, 2)
t_up, t_dn, t_both = run(True, False), run(False, True), run(True, True)
print(json.dumps({"bytes_each": n, "h2d_GBs": n / t_up / 1e9, "d2h_GBs": n / t_dn / 1e9, "both_GBs_each": n / t_both / 1e9,
                  "both_GBs_total": 2 * n / t_both / 1e9, "overlap": (t_up + t_dn) / t_both}))
