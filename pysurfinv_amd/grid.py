"""Grid-of-points inversion ("model3D" flow): many surface points, several
Metropolis chains per point, sharded over the GPUs of a node.

The reference has no driver for this: users run ``Point.MCinvMP`` per grid point and
``Model3D.loadInvDir`` (``model3D.py:36-57``) later reads ``invDir/{lon}_{lat}.npz`` and wraps each in a
``PostPoint`` (``point.py:134-175``).  Points and chains are fully independent, so ranks own contiguous blocks of
points (``shard.shard_range``), every rank advances ALL chains of its points in lock step (one batched forward solve
per Metropolis step), and the only collectives come after the sampling: an all-gather of one summary row per point
(what ``PostPoint`` derives: minimum-misfit and average accepted model, their misfits, the predicted curve -
``MetropolisBatch.summarise_points``) and the counter / timing reductions of ``shard.reduce_report`` (RCCL over xGMI
on GPUs, gloo in the CPU tests).  Each rank writes its own ``{lon}_{lat}.npz`` files with the reference's keys, so the
reference's post-processing can consume the directory unchanged - from writer threads, off the timed path.
"""
from __future__ import annotations

import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import shard
from .mcmc import MetropolisBatch

SUMMARY_HEAD = ("min_misfit", "min_L", "thres", "n_accepted_final", "avg_misfit", "avg_L")


def summary_columns(n_params, n_periods):
    """Column names of the gathered per-point summary rows."""
    return (list(SUMMARY_HEAD) + [f"min_p{i}" for i in range(n_params)] + [f"avg_p{i}" for i in range(n_params)]
            + [f"pvelp{i}" for i in range(n_periods)])


def run_grid(model_batch, lons, lats, periods, c_obs, uncer, chains_per_point, chainL, outdir=None,
             rank=0, world=1, device="cuda:0", seed=0, forward=None, isgood=None, fast_scan=False,
             writer_threads=4, keep_tracks=True, local_info=None, chain_groups=None, spec_depth=None):
    """Invert the points owned by ``rank``.

    model_batch : layers_batch.Model1DBatch (one setting for the grid)
    local_info  : None, or [n_points, K] - every point's own constants (``Point(setting, localInfo)``, point.py:8-14:
                  ``Info.topo`` / ``lithoAge`` / ``period``, fixed thicknesses ...), columns = ``model_batch.aux_names``
                  (the ``local_keys`` the model was built with).  The rank's block goes to the device once
                  (``Model1DBatch.set_local_info``); every chain reads its point's row.
    c_obs, uncer: [n_points, P] (NaN / non-positive uncertainty = masked period)
    chain_groups: None (the sampler's default: two groups of chains on two streams from 4 096 chains per rank on,
                  ``MetropolisBatch.chain_groups``) or their number; the random numbers of every chain do not depend on it
    spec_depth  : None (the sampler's default: speculative lock steps of depth 4 / 3 / 2 for up to 136 / 292 / 682 chains per rank,
                  ``MetropolisBatch.auto_spec_depth``) or the depth (1 = one step per forward solve)
    Returns dict(points=(lo, hi), mcTrack=[n_local, chains*chainL, 3+N] or None, summaries=[n_points, 6+2N+P]
    (every rank holds all rows, point order), columns, elapsed (sampling + summaries + gather, this rank),
    elapsed_write, report).  ``report`` carries the MAX over ranks of ``elapsed`` and the summed counters."""
    import torch
    lons, lats = np.asarray(lons), np.asarray(lats)
    c_obs, uncer = np.asarray(c_obs, float), np.asarray(uncer, float)
    n_points = lons.size
    lo, hi = shard.shard_range(n_points, rank, world)
    n_local = hi - lo
    C = n_local * chains_per_point
    N, P = model_batch.spec.n, len(periods)
    dev = torch.device(device)
    cdev = dev if dev.type != "cpu" else None
    t0 = time.perf_counter()
    n_forward = 0
    tracks_dev = torch.zeros((n_local, chains_per_point * chainL, 3 + N), dtype=torch.float64, device=dev)
    summ = torch.zeros((n_local, 6 + 2 * N + P), dtype=torch.float64, device=dev)
    if model_batch.n_aux and local_info is None:
        raise ValueError(f"the model has per-point constants {model_batch.aux_names}: pass local_info[n_points, {model_batch.n_aux}]")
    if n_local > 0:
        # chain index = point-major: chains of one point are consecutive
        rep = lambda a: np.repeat(a[lo:hi], chains_per_point, axis=0)
        local_rows = None
        if local_info is not None:
            li = np.asarray(local_info, float)
            if li.shape != (n_points, model_batch.n_aux):
                raise ValueError(f"local_info must be [{n_points}, {model_batch.n_aux}] (columns {model_batch.aux_names})")
            model_batch.set_local_info(li[lo:hi])                        # this rank's points only
            local_rows = np.repeat(np.arange(n_local), chains_per_point)
        mc = MetropolisBatch(model_batch.spec, model_batch.to_model, periods, rep(c_obs), rep(uncer),
                             device=device, seed=seed + 7919 * rank, forward=forward, isgood=isgood, fast_scan=fast_scan,
                             local_rows=local_rows)
        tracks_dev = mc.run_points(n_local, chains_per_point, chainL, on_device=True, groups=chain_groups, spec_depth=spec_depth).reshape(n_local, chains_per_point * chainL, -1)
        first_chain = torch.arange(n_local, device=dev) * chains_per_point     # observation row of each point
        summ = mc.summarise_points(tracks_dev, first_chain)
        n_forward = mc.n_forward
    # the one data exchange of the flow: every rank gets every point's summary row (a few hundred bytes per point)
    parts = shard.gather_rows(summ, device=cdev)
    summaries = torch.cat(parts, dim=0).cpu().numpy()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    # ---- outside the timed path: tracks to the host and {lon}_{lat}.npz files from writer threads
    t1 = time.perf_counter()
    tracks = tracks_dev.cpu().numpy() if (keep_tracks or outdir is not None) else None
    if outdir is not None and n_local > 0:
        def write(i):
            pid = f"{lons[lo + i]}_{lats[lo + i]}"              # model3D.py:41-47 file naming
            obs = {"T": list(np.asarray(periods, float)), "c": list(c_obs[lo + i]), "uncer": list(uncer[lo + i])}
            setting = model_batch.setting
            if local_info is not None:                          # the point's own setting, as Point(setting, localInfo) keeps it
                setting = model_batch.setting_for_row(np.asarray(local_info, float)[lo + i])
            return MetropolisBatch.save_npz(outdir, pid, tracks[i], setting, obs, chainL)
        with ThreadPoolExecutor(max_workers=max(1, int(writer_threads))) as ex:
            list(ex.map(write, range(n_local)))
    elapsed_write = time.perf_counter() - t1
    max_elapsed, (tot_forward, tot_points) = shard.reduce_report(elapsed, [n_forward, n_local], device=cdev)
    steps = tot_points * chains_per_point * chainL
    return dict(points=(lo, hi), mcTrack=tracks if keep_tracks else None, summaries=summaries,
                columns=summary_columns(N, P), elapsed=elapsed, elapsed_write=elapsed_write,
                report=dict(elapsed_max=max_elapsed, forward_solves=tot_forward, points=tot_points,
                            metropolis_steps=steps,
                            steps_per_s=steps / max_elapsed if max_elapsed > 0 else 0.0,
                            solves_per_s=tot_forward / max_elapsed if max_elapsed > 0 else 0.0))
