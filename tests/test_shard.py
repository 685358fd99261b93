"""N>1 path on CPU: world_size-2 gloo ranks exercise the sharding + reduction helpers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pysurfinv_amd import shard


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 4096, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_units, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_units, rank, world)
    # stand-in for the per-rank solve: a deterministic function of the unit index
    rows = torch.stack([torch.arange(lo, hi, dtype=torch.float32),
                        torch.arange(lo, hi, dtype=torch.float32) ** 2], dim=1)
    elapsed, (nsolved, nfail) = shard.reduce_report(0.1 * (rank + 1), [hi - lo, rank])
    allrows = torch.cat(shard.gather_rows(rows), dim=0)
    if rank == 0:
        q.put((elapsed, nsolved, nfail, allrows.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_report_and_gather():
    world, n_units = 2, 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs: p.start()
    elapsed, nsolved, nfail, rows = q.get(timeout=120)
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert abs(elapsed - 0.2) < 1e-12          # MAX over ranks
    assert nsolved == n_units and nfail == 1   # SUM over ranks
    assert rows.shape == (n_units, 2)
    assert np.array_equal(rows[:, 0], np.arange(n_units, dtype=np.float32))
