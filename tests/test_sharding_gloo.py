"""N>1 path on CPU: two processes over gloo exercise the sharding, max-over-ranks timing and final gather that
bench.py uses on RCCL.  The product kernels need a GPU, so the CPU oracle stands in as the per-shard compute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge

pkg = ge.load_package()
sh = pkg.sharding


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 7, 8, 1 << 20, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sh.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert len({sh.shard_seed(1, r) for r in range(8)}) == 8


def _worker(rank, world, port, q_out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ntru_oracle as orc
        N, q, d, total = 17, 32, 2, 10
        rng = np.random.default_rng(3)                      # same global batch on every rank
        h = rng.integers(0, q, N)
        r = np.zeros((total, N), np.uint8)
        for b in range(total):
            perm = rng.permutation(N); r[b, perm[:d]] = 1; r[b, perm[d:2 * d]] = 2
        m = rng.integers(0, 2, (total, N), dtype=np.uint8)
        lo, hi = sh.shard_range(total, rank, world)
        e_local, _ = orc.encrypt_batch(N, q, h, r[lo:hi], m[lo:hi])
        elapsed = sh.timed_region(lambda: None, lambda: None, dist, None)
        slow = sh.max_over_ranks(1.0 + rank, dist, None)
        allrows = sh.gather_rows(torch.from_numpy(e_local.astype(np.int32)), dist)
        if rank == 0:
            e_full, _ = orc.encrypt_batch(N, q, h, r, m)
            q_out.put((bool(np.array_equal(allrows.numpy(), e_full)), slow, elapsed >= 0.0, (lo, hi)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather_matches_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q_out)) for r in range(2)]
    [p.start() for p in procs]
    res = q_out.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    same, slow, ok, span = res
    assert same and slow == 2.0 and ok and span == (0, 5)
