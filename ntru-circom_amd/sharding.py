"""Multi-GPU plumbing: batches of independent items shard across ranks with no data-path collective
(SURVEY.md 8e).  One process per GPU; torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the
CPU tests) is used only for the barrier, the max-over-ranks timing and the optional final gather of outputs."""
import time


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` items owned by `rank`: slices differ by at most one item."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_seed(base_seed, rank):
    """Seed of a rank's synthetic shard (weak scaling: every rank generates its own B items)."""
    return int(base_seed) + 7919 * int(rank)


def _active(dist, always):
    """Collectives run when there is more than one rank -- or when the caller insists (a one-rank process group still
    goes through the backend: the only way to execute RCCL on a box with a single GPU)."""
    return dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or always)


def max_over_ranks(seconds, dist=None, device=None, always=False):
    """The slowest rank's elapsed time (the job is done when the last shard is)."""
    if not _active(dist, always):
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def timed_region(run_steps, sync, dist=None, device=None, always=False, return_local=False):
    """barrier + sync, run, sync + barrier; returns max-over-ranks seconds (bench.py contract).  return_local: (max over ranks,
    THIS rank's seconds up to its own sync, before the closing barrier) -- the per-rank figure a scaling record needs to show a
    straggler (one throttled GPU, a node-level power budget) instead of only its effect on the maximum."""
    multi = _active(dist, always)
    sync()
    if multi:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run_steps()
    sync()
    local = time.perf_counter() - t0
    if multi:
        dist.barrier()
    sync()
    worst = max_over_ranks(time.perf_counter() - t0, dist, device, always)
    return (worst, local) if return_local else worst


def rank_reports(me, dist=None, always=False):
    """Every rank's own record (a dict with at least "rank" and "ms_per_step"), gathered onto every rank and sorted by rank, plus a
    summary that names the slowest rank: (records, summary).  Without a process group: ([me], summary of one)."""
    if _active(dist, always):
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, me)
    else:
        everyone = [me]
    everyone = sorted(everyone, key=lambda x: x["rank"])
    ms = [float(x["ms_per_step"]) for x in everyone]
    slow = max(range(len(ms)), key=lambda i: ms[i])
    summary = {"slowest_rank": everyone[slow]["rank"], "ms_per_step_max": ms[slow], "ms_per_step_min": min(ms),
               "spread": ms[slow] / min(ms) - 1.0 if min(ms) > 0 else None}
    return everyone, summary


def gather_rows(local, dist, always=False, root=0):
    """The final gather (BASELINE north_star: "RCCL over xGMI only for the final gather"): per-rank output rows
    ([b_rank, N] each, equal b_rank) travel ONCE, onto `root`, in rank order -- a gather-to-root, not an all_gather: every
    other rank only sends, so the receive volume is (world - 1) shards on one GPU instead of on all of them (root's inbound
    xGMI links bound it).  Returns the concatenation on `root`, None elsewhere; `local` itself without a process group."""
    import torch
    if not _active(dist, always):
        return local
    local = local.contiguous()
    world, rank = dist.get_world_size(), dist.get_rank()
    outs = [torch.empty_like(local) for _ in range(world)] if rank == root else None
    dist.gather(local, outs, dst=root)
    return torch.cat(outs, dim=0) if rank == root else None


def shard_checksums(local, dist, always=False):
    """One int64 checksum per rank (a position-weighted sum of the rank's rows), all_gathered: 8 bytes per rank, so root can
    check every gathered shard against what its owner computed without a second copy of the data."""
    import torch
    flat = local.reshape(-1)
    mine = torch.zeros(1, dtype=torch.int64, device=flat.device)
    step = 1 << 24                                             # bounded temporaries: a shard may be gigabytes
    for o in range(0, flat.numel(), step):
        part = flat[o:o + step].to(torch.int64)
        w = (torch.arange(o, o + part.numel(), device=flat.device, dtype=torch.int64) % 65521) + 1
        mine += (part * w).sum()
    if not _active(dist, always):
        return mine
    outs = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, mine)
    return torch.cat(outs)
