#!/usr/bin/env python3
"""Grid-of-points inversion ("model3D" flow) on the GPUs of one node, one rank per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        scripts/run_grid.py --input grid.npz --setting setting.json --outdir mcdata [--chains 50 --chainL 1000]
    python scripts/run_grid.py --synthetic 64 --outdir mcdata            # one GPU, synthetic observations

``--input``: npz with lons[n], lats[n], periods[P], c_obs[n, P], uncer[n, P] (NaN = masked period) and, with
``--local-keys topo,lithoAge,...``, local_info[n, K]: every point's own constants (what the reference passes as
``Point(setting, localInfo)``, point.py:8-14; columns in the order of ``--local-keys``).
``--setting``: the reference's model setting as JSON (``models.py:42-51``; default: the continental example of
``pysurfinv_amd.settings``).  Every rank inverts its block of points (``pysurfinv_amd.grid.run_grid``), writes
``{outdir}/{lon}_{lat}.npz`` with the reference's keys - what ``Model3D.loadInvDir`` (``model3D.py:36-57``) reads - and
rank 0 writes ``{outdir}/summaries.npz``: one row per point of what ``PostPoint`` derives (minimum-misfit and average
accepted model, misfits, predicted curve), gathered over RCCL.  ``--backend gloo --device cpu`` needs a ``forward``
callable and is for tests only (the product path has no CPU solver).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic points instead of --input")
    ap.add_argument("--setting")
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--chains", type=int, default=50)
    ap.add_argument("--chainL", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fast-scan", action="store_true", help="opt into the heuristic scan (SURFDISP_FASTSCAN)")
    ap.add_argument("--local-keys", default="", help="comma-separated per-point constants (topo, lithoAge, period, <Layer>.<key>): "
                                                     "columns of local_info[n, K] in --input")
    ap.add_argument("--chain-groups", type=int, default=None, help="chain groups per rank (default: 2 from 4 096 chains on)")
    ap.add_argument("--spec-depth", type=int, default=None, help="Metropolis steps per batched solve (default: 4 / 3 / 2 for up to "
                                                                 "136 / 292 / 682 chains per rank, else 1)")
    args = ap.parse_args()

    import torch
    from pysurfinv_amd import settings
    from pysurfinv_amd import grid
    from pysurfinv_amd.layers_batch import Model1DBatch

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_REHEARSAL=1 (development only, as in bench.py): all ranks share cuda:0 and rendezvous over gloo, to
    # exercise this entry with several ranks on a one-GPU box
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda:0" if rehearsal else f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
    setting = json.load(open(args.setting)) if args.setting else settings.MCMC_SETTING
    local_keys = [k for k in args.local_keys.split(",") if k]
    mb = Model1DBatch(setting, device=dev, local_keys=local_keys)
    local_info = None
    if args.input:
        z = np.load(args.input)
        lons, lats, periods, c_obs, uncer = z["lons"], z["lats"], z["periods"], z["c_obs"], z["uncer"]
        if local_keys:
            local_info = z["local_info"]
    else:
        n = max(1, args.synthetic)
        _, c_obs, uncer = settings.synthetic_observations(n, dev)
        periods = np.asarray(settings.MCMC_PERIODS, float)
        lons, lats = 230.0 + 0.5 * (np.arange(n) % 64), 40.0 + 0.5 * (np.arange(n) // 64)
    r = grid.run_grid(mb, lons, lats, periods, c_obs, uncer, args.chains, args.chainL, outdir=args.outdir,
                      rank=rank, world=world, device=str(dev), seed=args.seed, fast_scan=args.fast_scan, keep_tracks=False,
                      local_info=local_info, chain_groups=args.chain_groups, spec_depth=args.spec_depth)
    if rank == 0:
        os.makedirs(args.outdir, exist_ok=True)
        np.savez_compressed(os.path.join(args.outdir, "summaries.npz"), summaries=r["summaries"], columns=np.array(r["columns"]),
                            lons=lons, lats=lats, periods=periods)
        rep = dict(r["report"]); rep.update(n_gpus=world, elapsed_write_rank0=r["elapsed_write"])
        print(json.dumps(rep), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
