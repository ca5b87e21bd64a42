# diagnostic: where do path-7 encrypt results differ from path 4 at the bench's size (device arrays)?
import importlib, numpy as np, sys, torch
sys.path.insert(0, '.')
pkg = importlib.import_module('ntru-circom_amd')
dev = torch.device('cuda:0')
eng = pkg.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
N, q = 821, 4096
for log2 in (16, 18, 20):
    B = (1 << log2) + (0 if log2 == 16 else 45)
    g = torch.Generator(device=dev); g.manual_seed(1)
    r = torch.randint(0, 3, (B, N), dtype=torch.uint8, device=dev, generator=g)
    m = torch.randint(0, 2, (B, N), dtype=torch.uint8, device=dev, generator=g)
    h = torch.randint(0, q, (N,), dtype=torch.int16, device=dev, generator=g)
    outs = {}
    ALT = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    for path in (4, ALT, ALT):
        e = torch.full((B, N), 0x5A5A, dtype=torch.int16, device=dev); qe = torch.full((B, N), 0x5A5A, dtype=torch.int16, device=dev)
        eng.set_kernel_path(path)
        eng.encrypt_batch_dev(N, q, h.data_ptr(), r.data_ptr(), m.data_ptr(), B, e.data_ptr(), qe.data_ptr())
        torch.cuda.synchronize()
        outs.setdefault(path, []).append((e, qe))
    for idx, name in ((0, "e"), (1, "quotE")):
        a, b, c = outs[4][0][idx], outs[ALT][0][idx], outs[ALT][1][idx]
        bad = (a != b).nonzero()
        print(log2, name, "mismatches 4 vs 7:", len(bad), " 7 vs 7 again:", int((b != c).sum()))
        if len(bad):
            rows = torch.unique(bad[:, 0]); cols = torch.unique(bad[:, 1])
            print(" rows", len(rows), rows[:16].tolist(), "row%32", torch.unique(rows % 32).tolist()[:40])
            print(" rowblocks", torch.unique(rows // 32)[:24].tolist(), "... max", int(rows.max()) // 32)
            print(" cols", len(cols), cols[:48].tolist())
            for (i, j) in bad[:10].tolist(): print("  ", i, j, hex(a[i, j].item() & 0xFFFF), hex(b[i, j].item() & 0xFFFF))
