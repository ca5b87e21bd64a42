# Socket power / shader clock while ONE kernel runs back to back for a few seconds (hwmon sampling as in bench.py):
#   python tools/power_kernel.py encrypt|decrypt|verify PATH [PATH ...]
#   python tools/power_kernel.py decrypt-operands 0      the UNCHANGED decrypt kernel on operands of different entropy: which operand carries the
#                                                        energy a matrix instruction costs above its bare price (results are not checked here)
import importlib, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module('ntru-circom_amd')
dev = torch.device('cuda:0')
eng = pkg.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
what = sys.argv[1]
N, q, p, B = 821, 4096, 3, 1 << 20
g = torch.Generator(device=dev); g.manual_seed(1)
r = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev, generator=g)
m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev, generator=g)
h = torch.randint(0, q, (N,), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
f = (torch.randint(0, 3, (N,), device=dev, generator=g) - 1).to(torch.int8)
fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev, generator=g)
e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev, generator=g).to(torch.int16); qe = torch.empty_like(e)
v = torch.empty((B, N), dtype=torch.uint8, device=dev); q2 = torch.empty_like(v); q1 = torch.empty_like(e); r1 = torch.empty_like(e)
e2 = torch.empty_like(e)
if what == 'verify':                                       # verifyKeysInputs on 2^18 synthetic per-item key pairs
    Bk = 1 << 18
    kf = (torch.randint(0, 3, (Bk, N), device=dev, generator=g) - 1).to(torch.int8); kg = (torch.randint(0, 3, (Bk, N), device=dev, generator=g) - 1).to(torch.int8)
    kfq = e[:Bk].contiguous(); kfp = r[:Bk].contiguous(); kh = qe[:Bk].random_(0, q)
    o16 = [torch.empty((Bk, N), dtype=torch.int16, device=dev) for _ in range(4)]; o8 = [torch.empty((Bk, N), dtype=torch.uint8, device=dev) for _ in range(2)]
    fl = torch.empty(Bk, dtype=torch.uint8, device=dev)
def call():
    if what == 'verify':
        eng.verify_keys_batch_dev(N, q, p, kf.data_ptr(), kg.data_ptr(), kfq.data_ptr(), kfp.data_ptr(), kh.data_ptr(), Bk, o16[0].data_ptr(), o16[1].data_ptr(),
                                  o8[0].data_ptr(), o8[1].data_ptr(), o16[2].data_ptr(), o16[3].data_ptr(), fl.data_ptr())
    elif what == 'encrypt':
        eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e2.data_ptr(), qe.data_ptr())
    else:
        eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr(), q1.data_ptr(), r1.data_ptr(), q2.data_ptr())
variants = [("as measured", None)]
if what == 'decrypt-operands':
    what = 'decrypt'
    rnd = e.clone()
    fz = torch.zeros_like(f); fone = torch.ones_like(f)
    variants = [("e uniform mod q, golden f (the benchmark's operands)", lambda: (e.copy_(rnd), None)),
                ("e = 0 (both digit planes constant zero)", lambda: (e.zero_(), None)),
                ("e = 1365 everywhere (constant non-zero digits)", lambda: (e.fill_(1365), None)),
                ("e < 128 (high digit plane zero, low plane random)", lambda: (e.copy_(rnd & 127), None)),
                ("e = 128 * (e >> 7) (low digit plane zero, high plane random)", lambda: (e.copy_(rnd & ~127), None)),
                ("e uniform, f = 0 (Toeplitz fragments all zero)", lambda: (e.copy_(rnd), f.copy_(fz))),
                ("e uniform, f = 1 everywhere (fragments constant)", lambda: (e.copy_(rnd), f.copy_(fone)))]
    f_saved = f.clone()
for label, prep in variants:
  if prep:
    f.copy_(f_saved); prep(); torch.cuda.synchronize(); print('--', label, flush=True)
  for path in [int(x) for x in sys.argv[2:]]:
    eng.set_kernel_path(path)
    for _ in range(3): call()
    torch.cuda.synchronize()
    sampler = bench.PowerSampler(bench.device_sysfs_dir(0)); sampler.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3.0:
        for _ in range(50): call(); n += 1
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    pw = sampler.summary(t0, t1)
    print('path %2d %-14s %.3f ms  %.0f W  %.0f MHz  -> %.2f J per launch (%.2f J above 300 W)' % (path, eng.last_kernel(), (t1 - t0) / n * 1e3, pw['socket_W'], pw['sclk_MHz'],
          pw['socket_W'] * (t1 - t0) / n, (pw['socket_W'] - 300) * (t1 - t0) / n), flush=True)
