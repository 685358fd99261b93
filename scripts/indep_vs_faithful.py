#!/usr/bin/env python3
"""GPU box: how far the (stack, period) decomposition (SURFDISP_INDEPENDENT) sits from the faithful period walk on
prior draws of the Metropolis models (monotone and not), phase only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import _lib, forward, settings
from pysurfinv_amd.brownian import TorchProposer
from pysurfinv_amd.layers_batch import Model1DBatch
dev = torch.device("cuda:0")
for name, setting, periods in (("continental 96 layers", settings.MCMC_SETTING, settings.MCMC_PERIODS),
                               ("thermal oceanic", settings.C5_SETTING, list(np.linspace(8, 100, 20)))):
    mb = Model1DBatch(setting, device=dev)
    p = TorchProposer(mb.spec, dev, seed=2).reset(20000)
    model, nlay = mb.to_model(p)
    per = torch.as_tensor(np.asarray(periods, np.float32), device=dev)
    plan = forward.BatchPlan(model.shape[0], model.shape[2], per.numel(), device=dev)
    cf = plan.run(model, per, kind=2 | 0x10, nlay=nlay)[0].clone(); sf = plan.status.clone()
    ci = plan.run(model, per, kind=2 | 0x10, nlay=nlay, independent=True)[0].clone(); si = plan.status.clone()
    torch.cuda.synchronize()
    vs = model[:, 1]
    L = model.shape[2]
    idx = torch.arange(L, device=dev)[None, :]
    nl = nlay if nlay is not None else torch.full((model.shape[0],), L, device=dev, dtype=torch.int32)
    valid = idx[:, 1:] < nl[:, None]
    mono = ((vs[:, 1:] >= vs[:, :-1]) | ~valid).all(dim=1) & ((model[:, 0, 1:] >= model[:, 0, :-1]) | ~valid).all(dim=1)
    same0 = ((cf > 0) == (ci > 0)).all(dim=1)
    both = (cf > 0) & (ci > 0)
    rel = torch.where(both, (ci.double() / cf.double().clamp(min=1e-9) - 1).abs(), torch.zeros_like(cf, dtype=torch.float64))
    for tag, sel in (("monotone", mono), ("not monotone", ~mono)):
        n = int(sel.sum())
        if n == 0:
            print(f"{name}: {tag}: none"); continue
        r = rel[sel]
        print(f"{name}: {tag}: {n} stacks; zero pattern differs on {int((~same0[sel]).sum())}; status differs on {int((sf[sel] != si[sel]).sum())}; "
              f"c rel diff max {float(r.max()):.2e} q99.9 {float(torch.quantile(r.flatten()[:2000000], 0.999)):.2e} median {float(r.flatten().median()):.2e}")
