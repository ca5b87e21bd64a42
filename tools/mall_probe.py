# Can a producer -> consumer hand-off of S megabytes between two kernels stay on the die (L2 / Infinity Cache) instead of going
# through HBM?  (The "decoupled" design for decryptBits: matrix-only kernels write raw tiles into a chunk-sized scratch, a streaming
# epilogue kernel reads them back.)  Per iteration: producer y = x_i (reads S fresh bytes, writes the scratch y), consumer z_i = y
# (reads the scratch, writes S fresh bytes).  "reused": ONE scratch of S bytes for every iteration; "fresh": a different S-byte
# region of a large buffer each time (what the hand-off costs when it goes through HBM).  Same kernels, same instruction counts.
import sys, torch
dev = torch.device('cuda:0')
BIG = 6 << 30
xs = torch.empty(BIG, dtype=torch.uint8, device=dev); xs.random_(0, 255)
zs = torch.empty(BIG, dtype=torch.uint8, device=dev)
ys = torch.empty(BIG, dtype=torch.uint8, device=dev)
def run(S, reused, iters):
    n = BIG // S
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        if rep == 1: ev0.record()
        for i in range(iters):
            k = i % n
            y = ys[:S] if reused else ys[k * S:(k + 1) * S]
            y.copy_(xs[k * S:(k + 1) * S])
            zs[k * S:(k + 1) * S].copy_(y)
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters
print('S MB | reused scratch: ms, GB/s of the 4 S bytes moved | fresh scratch: ms, GB/s | reused / fresh')
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    S = mb << 20
    iters = max(8, min(400, (24 << 30) // S))
    a = run(S, True, iters); b = run(S, False, iters)
    print('%5d | %.4f ms %7.0f | %.4f ms %7.0f | %.2f' % (mb, a, 4 * S / a / 1e6, b, 4 * S / b / 1e6, a / b), flush=True)
