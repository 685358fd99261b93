#!/usr/bin/env python3
"""GPU box: how many prior draws of the bench's Metropolis models the production root search hands to the exact
fallback kernel, and what a lock step's forward solve costs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import settings as bench
from pysurfinv_amd import forward, _lib
from pysurfinv_amd.layers_batch import Model1DBatch
from pysurfinv_amd.brownian import TorchProposer
dev = torch.device("cuda:0")
for name, setting, periods in (("continental 96 layers", bench.MCMC_SETTING, bench.MCMC_PERIODS),):
    mb = Model1DBatch(setting, device=dev)
    for C in (100, 25600):
        p = TorchProposer(mb.spec, dev, seed=1).reset(C)
        model, nlay = mb.to_model(p)
        per = torch.as_tensor(np.asarray(periods, np.float32), device=dev)
        plan = forward.BatchPlan(C, model.shape[2], per.numel(), device=dev)
        for kind, kn in ((2 | 0x10, "phase only"), (2, "c+U")):
            plan.run(model, per, kind=kind, nlay=nlay); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): plan.run(model, per, kind=kind, nlay=nlay)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            c, u, st, ms = plan.run_timed(model, per, kind=kind, nlay=nlay)
            print(f"{name}, {C} chains, {kn}: team {_lib.lib().surfdisp_get_team(C, model.shape[2])}, {dt * 1e3:.2f} ms per solve, "
                  f"kernels prep/phase(+fallback)/group {ms[0]:.3f}/{ms[1]:.3f}/{ms[2]:.3f} ms, "
                  f"{plan.fallback_count()} stacks through the exact fallback, status!=0: {int((st != 0).sum())}")
