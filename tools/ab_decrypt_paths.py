# same-device timing and equality of the decrypt kernels by kernel path (device arrays, HIP events), witness and value-only
import importlib, sys, torch
sys.path.insert(0, '.')
pkg = importlib.import_module('ntru-circom_amd')
dev = torch.device('cuda:0')
eng = pkg.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
p = 3
for N, q in ((509, 2048), (701, 8192), (821, 4096), (167, 128)):
    B = (1 << 20) + 13
    g = torch.Generator(device=dev); g.manual_seed(N)
    e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
    f = (torch.randint(0, 3, (N,), dtype=torch.int8, device=dev, generator=g) - 1).to(torch.int8)
    fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev, generator=g)
    ref = None; out = {}
    for rnd in range(2):
        for path in (4, 5, 8):
            eng.set_kernel_path(path)
            v = torch.empty((B, N), dtype=torch.uint8, device=dev); q2 = torch.empty_like(v)
            q1 = torch.empty((B, N), dtype=torch.int16, device=dev); r1 = torch.empty_like(q1)
            args = (N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr(), q1.data_ptr(), r1.data_ptr(), q2.data_ptr())
            for _ in range(2): eng.decrypt_batch_dev(*args)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5): eng.decrypt_batch_dev(*args)
            t1.record(); torch.cuda.synchronize()
            name = eng.last_kernel()
            if ref is None: ref = (v, q1, r1, q2)
            same = all(torch.equal(a, b) for a, b in zip(ref, (v, q1, r1, q2)))
            out.setdefault((path, name), []).append((round(t0.elapsed_time(t1) / 5, 3), same))
    print(N, q, {"%d %s" % k: v for k, v in out.items()})
