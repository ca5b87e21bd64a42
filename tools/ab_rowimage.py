# Experimental kernel paths (AB_NEW_PATH: 10 = row-image kernels of matrix_rowimage.hip [default], 11 = decrypt with the fp4 second
# product) against the default matrix kernels: bit-exactness on ragged batches and
# misaligned arrays, then same-device timing at 2^20 items (HIP events, interleaved rounds).
#   python tools/ab_rowimage.py [check|time|all] [encrypt|decrypt|both]
import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module('ntru-circom_amd')
dev = torch.device('cuda:0')
eng = pkg.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
which = sys.argv[2] if len(sys.argv) > 2 else 'both'
import os
NEW = int(os.environ.get('AB_NEW_PATH', '10'))
DEC_NAMES = ('k_decrypt_w',) if NEW == 10 else ('k_decrypt_m8q', 'k_decrypt_mq')


def carve(nbytes_list, off):
    """Tensors that start `off` bytes into fresh 16-byte aligned allocations."""
    outs = []
    for nb in nbytes_list:
        buf = torch.zeros(nb + 64, dtype=torch.uint8, device=dev)
        outs.append(buf[off:off + nb])
    return outs


def check_encrypt():
    bad = 0
    for N, q in ((821, 4096), (701, 8192), (509, 2048), (167, 128), (64, 16), (255, 512), (257, 1024), (800, 4096)):
        for B in (1, 31, 32, 33, 500, 8195, 32 * 256 * 3 + 5):
            for off in (0, 2, 6, 14):
                g = torch.Generator(device=dev); g.manual_seed(N * 7 + B + off)
                r = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev, generator=g)
                m = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev, generator=g)
                h = torch.randint(0, q, (N,), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
                moff = (off * 3 + 1) % 16
                rb, mb = carve([B * N, B * N], moff)
                rb.copy_(r.view(-1)); mb.copy_(m.view(-1))
                res = {}
                for path in (4, NEW):
                    eb, qb = carve([2 * B * N, 2 * B * N], off)
                    guard_e = eb.storage() if False else None
                    eng.set_kernel_path(path)
                    eng.encrypt_batch_dev(N, q, h.data_ptr(), rb.data_ptr(), mb.data_ptr(), B, eb.data_ptr(), qb.data_ptr())
                    torch.cuda.synchronize()
                    res[path] = (eb.clone(), qb.clone(), eng.last_kernel())
                    # value-only mode as well
                    eb2, = carve([2 * B * N], off)
                    eng.encrypt_batch_dev(N, q, h.data_ptr(), rb.data_ptr(), mb.data_ptr(), B, eb2.data_ptr(), None)
                    torch.cuda.synchronize()
                    if not torch.equal(eb2, eb):
                        print('MISMATCH value-only', N, q, B, off, path); bad += 1
                same = torch.equal(res[4][0], res[NEW][0]) and torch.equal(res[4][1], res[NEW][1])
                if not same or res[NEW][2] != 'k_encrypt_w':
                    bad += 1
                    de = (res[4][0] != res[NEW][0]).nonzero().flatten()
                    dq = (res[4][1] != res[NEW][1]).nonzero().flatten()
                    print('MISMATCH encrypt', N, q, B, off, res[NEW][2], 'e diffs', de.numel(), de[:6].tolist(), 'q diffs', dq.numel(), dq[:6].tolist())
    print('encrypt check: %s' % ('OK' if bad == 0 else '%d FAILURES' % bad))
    return bad


def time_encrypt():
    import os
    for N, q in ((821, 4096),) if os.environ.get('RI_ONLY821') else ((821, 4096), (701, 8192), (509, 2048)):
        B = 1 << 20
        g = torch.Generator(device=dev); g.manual_seed(N)
        r = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev, generator=g)
        m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev, generator=g)
        h = torch.randint(0, q, (N,), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
        e = torch.empty((B, N), dtype=torch.int16, device=dev); qe = torch.empty_like(e)
        out = {}
        for rnd in range(3):
            for path in (0, NEW):
                for wit in (True, False):
                    eng.set_kernel_path(path)
                    args = (N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), qe.data_ptr() if wit else None)
                    for _ in range(2): eng.encrypt_batch_dev(*args)
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    for _ in range(10): eng.encrypt_batch_dev(*args)
                    t1.record(); torch.cuda.synchronize()
                    out.setdefault((eng.last_kernel(), wit), []).append(round(t0.elapsed_time(t1) / 10, 3))
        print(N, q, {k[0] + (' witness' if k[1] else ' value-only'): v for k, v in out.items()}, flush=True)


def check_decrypt():
    bad = 0
    p = 3
    for N, q in ((821, 4096), (701, 8192), (509, 2048), (167, 128), (64, 16), (255, 512), (257, 1024), (800, 4096)):
        for B in (1, 31, 32, 33, 500, 8195):
            for off in (0, 2, 6, 14):
                g = torch.Generator(device=dev); g.manual_seed(N * 7 + B + off)
                e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
                f = (torch.randint(0, 3, (N,), device=dev, generator=g) - 1).to(torch.int8)
                fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev, generator=g)
                eb, = carve([2 * B * N], off)
                eb.copy_(e.view(torch.uint8).view(-1))
                res = {}
                for path in (4, NEW):
                    vb, q2b = carve([B * N, B * N], (off * 5 + 3) % 16)
                    q1b, r1b = carve([2 * B * N, 2 * B * N], off)
                    eng.set_kernel_path(path)
                    eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), eb.data_ptr(), B, vb.data_ptr(), q1b.data_ptr(), r1b.data_ptr(), q2b.data_ptr())
                    torch.cuda.synchronize()
                    res[path] = (vb.clone(), q1b.clone(), r1b.clone(), q2b.clone(), eng.last_kernel())
                    vb2, = carve([B * N], (off * 5 + 3) % 16)
                    eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), eb.data_ptr(), B, vb2.data_ptr(), None, None, None)
                    torch.cuda.synchronize()
                    if not torch.equal(vb2, vb):
                        print('MISMATCH value-only', N, q, B, off, path); bad += 1
                same = all(torch.equal(res[4][i], res[NEW][i]) for i in range(4))
                if not same or res[NEW][4] not in DEC_NAMES:
                    bad += 1
                    print('MISMATCH decrypt', N, q, B, off, res[NEW][4],
                          [int((res[4][i] != res[NEW][i]).sum()) for i in range(4)],
                          [(res[4][i] != res[NEW][i]).nonzero().flatten()[:4].tolist() for i in range(4)])
    print('decrypt check: %s' % ('OK' if bad == 0 else '%d FAILURES' % bad))
    return bad


def time_decrypt():
    p = 3
    for N, q in ((821, 4096), (701, 8192), (509, 2048)):
        B = 1 << 20
        g = torch.Generator(device=dev); g.manual_seed(N)
        e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
        f = (torch.randint(0, 3, (N,), device=dev, generator=g) - 1).to(torch.int8)
        fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev, generator=g)
        v = torch.empty((B, N), dtype=torch.uint8, device=dev); q2 = torch.empty_like(v)
        q1 = torch.empty((B, N), dtype=torch.int16, device=dev); r1 = torch.empty_like(q1)
        out = {}
        for rnd in range(3):
            for path in (0, NEW):
                for wit in (True, False):
                    eng.set_kernel_path(path)
                    args = (N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr()) + ((q1.data_ptr(), r1.data_ptr(), q2.data_ptr()) if wit else (None, None, None))
                    for _ in range(2): eng.decrypt_batch_dev(*args)
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    for _ in range(10): eng.decrypt_batch_dev(*args)
                    t1.record(); torch.cuda.synchronize()
                    out.setdefault((eng.last_kernel(), wit), []).append(round(t0.elapsed_time(t1) / 10, 3))
        print(N, q, {k[0] + (' witness' if k[1] else ' value-only'): v for k, v in out.items()}, flush=True)


rc = 0
if what in ('check', 'all'):
    if which in ('encrypt', 'both'): rc += check_encrypt()
    if which in ('decrypt', 'both'): rc += check_decrypt()
if what in ('time', 'all') and rc == 0:
    if which in ('encrypt', 'both'): time_encrypt()
    if which in ('decrypt', 'both'): time_decrypt()
sys.exit(1 if rc else 0)
