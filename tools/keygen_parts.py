"""Key inversion by parts on the GPU: mod-p inversion alone (k_invert_key<3>), the fq chain alone (k_invert_key<2> + Newton rounds),
both (ntru_invert_key_batch_dev as key generation calls it).     python tools/keygen_parts.py [logB]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from bench_configs import eng, dev, timed          # noqa: E402  (the engine handle and the timer the config lines use)

logB = int(sys.argv[1]) if len(sys.argv) > 1 else 18
o, _, _, _ = bench.load_key("n821_q4096")
N, q, p, df = o["N"], o["q"], o["p"], o["df"]
B = 1 << logB
key = (np.arange(8, dtype=np.uint32) * 0x85EBCA6B + 7).astype(np.uint32)
fs = torch.empty((B, N), dtype=torch.uint8, device=dev)
eng.sample_ternary_dev(N, df, df - 1, 255, key, 0, B, fs.data_ptr())
f = fs.view(torch.int8)
fq = torch.empty((B, N), dtype=torch.int16, device=dev); fp = torch.empty((B, N), dtype=torch.uint8, device=dev)
fl = torch.empty(B, dtype=torch.uint8, device=dev)
out = {"config": "N=%d q=%d key inversion of 2^%d sampled f" % (N, q, logB)}
out["modp_only_ms"] = timed(lambda: eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, None, fp.data_ptr(), fl.data_ptr()), steps=3, warmup=1)
out["fq_chain_only_ms"] = timed(lambda: eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, fq.data_ptr(), None, fl.data_ptr()), steps=3, warmup=1)
out["mod2_only_ms"] = timed(lambda: eng.invert_key_batch_dev(N, 2, p, f.data_ptr(), B, fq.data_ptr(), None, fl.data_ptr()), steps=3, warmup=1)
out["both_ms"] = timed(lambda: eng.invert_key_batch_dev(N, q, p, f.data_ptr(), B, fq.data_ptr(), fp.data_ptr(), fl.data_ptr()), steps=3, warmup=1)
out["units"] = int((fl == 0).sum())
print(json.dumps(out))
