import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module('ntru-circom_amd')
from oracle import ntru_oracle as orc
dev = torch.device('cuda:0')
eng = pkg.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
bad = 0
p = 3
for N, q in ((821, 4096), (701, 8192), (509, 2048), (126, 64), (127, 64), (252, 256), (253, 256), (378, 512), (1000, 1024)):
    for B in (1, 31, 33, 500, 8195):
        g = torch.Generator(device=dev); g.manual_seed(N + B)
        e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
        f = (torch.randint(0, 3, (N,), device=dev, generator=g) - 1).to(torch.int8)
        fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev, generator=g)
        bits, per, al, os_ = eng.pack_params(2, N) if hasattr(eng, 'pack_params') else (None,)*4
        v_ref = torch.empty((B, N), dtype=torch.uint8, device=dev)
        eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v_ref.data_ptr())
        torch.cuda.synchronize()
        want = orc.pack_batch(2, N, v_ref.cpu().numpy().astype(np.uint16))
        os_ = want.shape[1]
        for with_value in (True, False):
            packed = torch.full((B, os_, 4), -1, dtype=torch.int64, device=dev)
            v = torch.full((B, N), 7, dtype=torch.uint8, device=dev)
            eng.decrypt_pack_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr() if with_value else None, packed.data_ptr())
            torch.cuda.synchronize()
            got = packed.cpu().numpy().view(np.uint64)
            ok = np.array_equal(got, want.view(np.uint64).reshape(got.shape)) and (not with_value or torch.equal(v, v_ref))
            if not ok:
                bad += 1
                d = np.argwhere(got != want.view(np.uint64).reshape(got.shape))
                print('MISMATCH', N, q, B, with_value, eng.last_kernel(), len(d), d[:4].tolist(), [hex(int(got[tuple(i)])) for i in d[:2]], [hex(int(want.view(np.uint64).reshape(got.shape)[tuple(i)])) for i in d[:2]])
        name = eng.last_kernel()
    print(N, q, name, 'bad so far', bad, flush=True)
print('fused pack:', 'OK' if not bad else '%d FAILURES' % bad)
# timing at 2^20
N, q, B = 821, 4096, 1 << 20
g = torch.Generator(device=dev); g.manual_seed(1)
e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
f = (torch.randint(0, 3, (N,), device=dev, generator=g) - 1).to(torch.int8)
fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev, generator=g)
v = torch.empty((B, N), dtype=torch.uint8, device=dev); packed = torch.empty((B, 7, 4), dtype=torch.int64, device=dev)
def t(fn, n=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rnd in range(3):
    t_sep = t(lambda: (eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr()), eng.pack_bytes_batch_dev(2, N, v.data_ptr(), B, packed.data_ptr())))
    t_fv = t(lambda: eng.decrypt_pack_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr(), packed.data_ptr()))
    t_f = t(lambda: eng.decrypt_pack_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, None, packed.data_ptr()))
    print('decrypt (value only) + k_pack: %.3f ms | fused, value + packed: %.3f ms | fused, packed only: %.3f ms' % (t_sep, t_fv, t_f), flush=True)
