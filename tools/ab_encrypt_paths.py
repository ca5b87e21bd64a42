# same-device timing of the encrypt kernels by kernel path at the secondary configurations (device arrays, HIP events)
import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module('ntru-circom_amd')
dev = torch.device('cuda:0')
eng = pkg.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for N, q in ((509, 2048), (701, 8192), (821, 4096), (167, 128), (1024, 8192)):
    B = 1 << 20
    g = torch.Generator(device=dev); g.manual_seed(N)
    r = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev, generator=g)
    m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev, generator=g)
    h = torch.randint(0, q, (N,), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
    e = torch.empty((B, N), dtype=torch.int16, device=dev); qe = torch.empty_like(e)
    out = {}
    for rnd in range(2):
        for path in (4, 5):
            for wit in (True, False):
                eng.set_kernel_path(path)
                args = (N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), qe.data_ptr() if wit else None)
                for _ in range(2): eng.encrypt_batch_dev(*args)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(5): eng.encrypt_batch_dev(*args)
                t1.record(); torch.cuda.synchronize()
                out.setdefault((eng.last_kernel(), wit), []).append(round(t0.elapsed_time(t1) / 5, 3))
    print(N, q, {k[0] + (' witness' if k[1] else ' value-only'): v for k, v in out.items()})
