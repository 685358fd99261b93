#!/usr/bin/env python3
"""GPU box: a single point's 100 chains (BASELINE configs[2]) - Metropolis steps/s of MetropolisBatch.run for the plain
lock step (fused kernels), the speculative sampler (spec_depth = 2, 3, 4: 2^d - 1 proposals per chain and lock step
through ONE batched solve, d steps per lock step, same chain distribution) and the (stack, period) decomposition."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysurfinv_amd import settings
from pysurfinv_amd.layers_batch import Model1DBatch
from pysurfinv_amd.mcmc import MetropolisBatch
dev = torch.device("cuda:0")
mb, c_obs, unc = settings.synthetic_observations(1, dev)
C, chainL = int(os.environ.get("CHAINS", "100")), 240
for name, kw, ctor in (("default (device kernels, auto depth)", {}, {}), ("one step per solve (spec_depth=1)", {"spec_depth": 1}, {}),
                       ("torch glue, one step per solve", {"fused": False}, {}), ("torch glue, speculative d=3", {"fused": False, "spec_depth": 3}, {}),
                       ("speculative d=2", {"spec_depth": 2}, {}), ("speculative d=3", {"spec_depth": 3}, {}), ("speculative d=4", {"spec_depth": 4}, {}),
                       ("independent=auto (default depth 1)", {}, {"independent": "auto"}),
                       ("independent=auto, speculative d=3", {"spec_depth": 3}, {"independent": "auto"})):
    mc = MetropolisBatch(mb.spec, mb.to_model, settings.MCMC_PERIODS, c_obs[0], unc[0], device=dev, seed=3, **ctor)
    mc.run(C, 24, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr = mc.run(C, chainL, **kw); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc = float(tr[:, 1:, 2].mean())
    print(f"{name:40s}: {C * chainL / dt / 1e3:8.1f} k steps/s  ({dt / chainL * 1e3:.3f} ms per step of {C} chains; accept rate {acc:.3f}; "
          f"{mc.n_forward / (C * chainL):.2f} forward solves per step)", flush=True)
