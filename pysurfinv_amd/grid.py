"""Grid-of-points inversion ("model3D" flow): many surface points, several
Metropolis chains per point, sharded over the GPUs of a node.

The reference has no driver for this: users run ``Point.MCinvMP`` per grid point and
``Model3D.loadInvDir`` (``model3D.py:36-57``) later reads ``invDir/{lon}_{lat}.npz``.  Points and
chains are fully independent, so ranks own contiguous blocks of points (``shard.shard_range``),
every rank advances ALL chains of its points in lock step (one batched forward solve per Metropolis
step), and the only collectives are the report reductions in ``shard.reduce_report`` (RCCL over
xGMI on GPUs, gloo in the CPU tests).  Each rank writes its own ``{lon}_{lat}.npz`` files with the
reference's keys, so the reference's post-processing can consume the directory unchanged.
"""
from __future__ import annotations

import time

import numpy as np

from . import shard
from .mcmc import MetropolisBatch


def run_grid(model_batch, lons, lats, periods, c_obs, uncer, chains_per_point, chainL, outdir=None,
             rank=0, world=1, device="cuda:0", seed=0, forward=None, isgood=None):
    """Invert the points owned by ``rank``.

    model_batch : layers_batch.Model1DBatch (shared setting; per-point priors are the caller's job)
    c_obs, uncer: [n_points, P] (NaN / non-positive uncertainty = masked period)
    Returns dict(points=(lo, hi), mcTrack=[n_local, chains*chainL, 3+N], elapsed, report)."""
    import torch
    lons, lats = np.asarray(lons), np.asarray(lats)
    c_obs, uncer = np.asarray(c_obs, float), np.asarray(uncer, float)
    n_points = lons.size
    lo, hi = shard.shard_range(n_points, rank, world)
    n_local = hi - lo
    C = n_local * chains_per_point
    t0 = time.perf_counter()
    tracks = np.zeros((n_local, chains_per_point * chainL, 3 + model_batch.spec.n))
    n_forward = 0
    if n_local > 0:
        # chain index = point-major: chains of one point are consecutive
        rep = lambda a: np.repeat(a[lo:hi], chains_per_point, axis=0)
        mc = MetropolisBatch(model_batch.spec, model_batch.to_model, periods, rep(c_obs), rep(uncer),
                             device=device, seed=seed + 7919 * rank, forward=forward, isgood=isgood)
        tr = mc.run_points(n_local, chains_per_point, chainL)
        tracks = tr.reshape(n_local, chains_per_point * chainL, -1)
        n_forward = mc.n_forward
        if outdir is not None:
            for i in range(n_local):
                pid = f"{lons[lo + i]}_{lats[lo + i]}"              # model3D.py:41-47 file naming
                obs = {"T": list(np.asarray(periods, float)), "c": list(c_obs[lo + i]), "uncer": list(uncer[lo + i])}
                MetropolisBatch.save_npz(outdir, pid, tracks[i], model_batch.setting, obs, chainL)
    elapsed = time.perf_counter() - t0
    dev = torch.device(device) if str(device) != "cpu" else None
    max_elapsed, (tot_forward, tot_points) = shard.reduce_report(elapsed, [n_forward, n_local], device=dev)
    return dict(points=(lo, hi), mcTrack=tracks, elapsed=elapsed,
                report=dict(elapsed_max=max_elapsed, forward_solves=tot_forward, points=tot_points,
                            solves_per_s=tot_forward / max_elapsed if max_elapsed > 0 else 0.0))
