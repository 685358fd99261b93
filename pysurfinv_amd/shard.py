"""Sharding of independent work units (stacks / grid points / chains) over ranks.

The forward path has no exchange step: every (grid point, chain) is independent
(point.py:101-107), so units are dealt to ranks in contiguous blocks and the only collectives are
a MAX-reduce of the elapsed time and a SUM-reduce of counters for the throughput report
(RCCL over xGMI on the GPUs; gloo in the CPU tests).
"""
from __future__ import annotations


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank; sizes differ by at most one, every unit owned once."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def reduce_report(elapsed_s: float, counters, device=None):
    """(max elapsed over ranks, summed counters) via torch.distributed if initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), [int(c) for c in counters]
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    c = torch.tensor(list(counters), dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [int(x) for x in c.tolist()]


def gather_rows(local_rows, device=None):
    """all_gather of per-unit summary rows (float32 [n_local, w]) -> list over ranks (rank order)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [local_rows]
    world = dist.get_world_size()
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    nmax = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((nmax, local_rows.shape[1]), dtype=local_rows.dtype, device=device)
    pad[: local_rows.shape[0]] = local_rows
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o[: int(s.item())] for o, s in zip(out, sizes)]
