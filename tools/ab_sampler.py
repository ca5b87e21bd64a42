#!/usr/bin/env python3
"""k_sample_ternary: ms per 2^20 items at N = 821 / 701 / 509 (HIP events, 5 launches after 2 warm-ups) + the first and last rows
against the oracle.  NTRU_ENGINE_LIB selects the build."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import ntru_oracle as orc
pkg = ge.load_package()
eng = pkg.Engine(0)
st = torch.cuda.current_stream(); eng.set_stream(st.cuda_stream)
key = np.arange(8, dtype=np.uint32) + 1
for N, d in ((821, 273), (701, 233), (509, 169)):
    B = 1 << 20
    r = torch.empty((B, N), dtype=torch.uint8, device="cuda:0")
    fn = lambda: eng.sample_ternary_dev(N, d, d, 2, key, 0, B, r.data_ptr())
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5): fn()
    e1.record(st); torch.cuda.synchronize()
    ok = bool(np.array_equal(r[:4].cpu().numpy(), orc.sample_ternary_batch(N, d, d, 2, key, 0, 4)) and
              np.array_equal(r[B - 4:].cpu().numpy(), orc.sample_ternary_batch(N, d, d, 2, key, B - 4, 4)))
    print(json.dumps({"lib": os.environ.get("NTRU_ENGINE_LIB", "in-tree"), "N": N, "ms_per_2^20": e0.elapsed_time(e1) / 5, "rows_equal_oracle": ok}))
