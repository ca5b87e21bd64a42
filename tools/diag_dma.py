# diagnostic: decrypt through kernel path 8 (k_decrypt_m8d, direct-to-LDS row loads) against path 5 and against itself at the bench's size
import importlib, numpy as np, sys, torch
sys.path.insert(0, '.')
pkg = importlib.import_module('ntru-circom_amd')
dev = torch.device('cuda:0')
eng = pkg.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
N, q, p = 821, 4096, 3
for log2 in (16, 20, 20):
    B = (1 << log2) + (0 if log2 == 16 else 77)
    g = torch.Generator(device=dev); g.manual_seed(log2)
    e = torch.randint(0, q, (B, N), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
    f = (torch.randint(0, 3, (N,), dtype=torch.int8, device=dev, generator=g) - 1).to(torch.int8)
    fp = torch.randint(0, 3, (N,), dtype=torch.uint8, device=dev, generator=g)
    outs = []
    for path in (5, 8, 8):
        v = torch.full((B, N), 0xA5, dtype=torch.uint8, device=dev); q2 = torch.full((B, N), 0xA5, dtype=torch.uint8, device=dev)
        q1 = torch.full((B, N), 0x5A5A, dtype=torch.int16, device=dev); r1 = torch.full((B, N), 0x5A5A, dtype=torch.int16, device=dev)
        eng.set_kernel_path(path)
        eng.decrypt_batch_dev(N, q, p, f.data_ptr(), fp.data_ptr(), e.data_ptr(), B, v.data_ptr(), q1.data_ptr(), r1.data_ptr(), q2.data_ptr())
        torch.cuda.synchronize()
        outs.append((eng.last_kernel(), v, q1, r1, q2))
    print(B, [o[0] for o in outs])
    for idx, name in ((1, "value"), (2, "quot1"), (3, "rem1"), (4, "quot2")):
        a, b, c = outs[0][idx], outs[1][idx], outs[2][idx]
        bad = (a != b).nonzero()
        print(" ", name, "mismatches 5 vs 8:", len(bad), " 8 vs 8 again:", int((b != c).sum()))
        if len(bad):
            rows = torch.unique(bad[:, 0]); cols = torch.unique(bad[:, 1])
            print("   rows", len(rows), rows[:16].tolist(), "row%32", torch.unique(rows % 32).tolist()[:40], "rowblocks", torch.unique(rows // 32)[:16].tolist())
            print("   cols", len(cols), cols[:32].tolist())
