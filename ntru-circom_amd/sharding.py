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


def timed_region(run_steps, sync, dist=None, device=None, always=False):
    """barrier + sync, run, sync + barrier; returns max-over-ranks seconds (bench.py contract)."""
    multi = _active(dist, always)
    sync()
    if multi:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run_steps()
    sync()
    if multi:
        dist.barrier()
    sync()
    return max_over_ranks(time.perf_counter() - t0, dist, device, always)


def gather_rows(local, dist, always=False):
    """Final gather of per-rank output rows ([b_rank, N] each, equal b_rank) onto every rank, in rank order."""
    import torch
    if not _active(dist, always):
        return local
    outs = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, local.contiguous())
    return torch.cat(outs, dim=0)
