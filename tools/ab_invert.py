#!/usr/bin/env python3
"""ntru_invert_key_batch_dev: ms per 2^18 keys at N = 821, q = 4096 (whole call: mod-2 inversion + Newton rounds + mod-3 inversion) and
the share of k_invert_key<3,.> alone (fq not asked for), HIP events; f * fq = 1 checked on a sample by the product kernel."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
eng = pkg.Engine(0)
st = torch.cuda.current_stream(); eng.set_stream(st.cuda_stream)
N, q, p, B, df = 821, 4096, 3, 1 << 18, 273
key = np.arange(8, dtype=np.uint32) + 3
fs = torch.empty((B, N), dtype=torch.uint8, device="cuda:0")
eng.sample_ternary_dev(N, df, df - 1, 255, key, 0, B, fs.data_ptr())
f = fs.view(torch.int8)
fq = torch.empty((B, N), dtype=torch.int16, device="cuda:0"); fp = torch.empty((B, N), dtype=torch.uint8, device="cuda:0")
fl = torch.empty(B, dtype=torch.uint8, device="cuda:0")
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
both = timed(lambda: eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, fq.data_ptr(), fp.data_ptr(), fl.data_ptr()))
only3 = timed(lambda: eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, None, fp.data_ptr(), fl.data_ptr()))
units = int((fl == 0).sum())
# f * fp = 1 mod 3 and f * fq = 1 mod q on the first 4096 keys (the per-item product kernel, oracle-tested elsewhere)
n = 4096
f16 = (f[:n].to(torch.int32) % q).to(torch.int16); f3 = (f[:n].to(torch.int32) % 3).to(torch.int16)
quot = torch.empty((n, N), dtype=torch.int16, device="cuda:0"); rem = torch.empty((n, N), dtype=torch.int16, device="cuda:0")
eng.polymul_split_dev(N, q, f16.data_ptr(), fq.data_ptr(), n, quot.data_ptr(), rem.data_ptr()); torch.cuda.synchronize()
ok_q = bool((rem[:, 0] == 1).all() and (rem[:, 1:] == 0).all())
eng.polymul_split_dev(N, 3, f3.data_ptr(), fp[:n].to(torch.int16).data_ptr(), n, quot.data_ptr(), rem.data_ptr()); torch.cuda.synchronize()
ok_p = bool((rem[:, 0] == 1).all() and (rem[:, 1:] == 0).all())
print(json.dumps({"lib": os.environ.get("NTRU_ENGINE_LIB", "in-tree"), "invert_fq_and_fp_ms_per_2^18": both, "invert_fp_only_ms": only3,
                  "units": units, "f_fq_is_1": ok_q, "f_fp_is_1": ok_p}))
