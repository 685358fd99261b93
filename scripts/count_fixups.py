#!/usr/bin/env python3
"""GPU: how often the root search's rare paths are taken on the bench workloads (surfdisp_workspace_counters): stacks through the
exact fallback kernel, brackets refined with NEVILL by the vertical-phase test, ellipticities evaluated again with the reference's
arithmetic - per (stack, period)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import forward, synth, _lib
from pysurfinv_amd.brownian import TorchProposer
from pysurfinv_amd.layers_batch import Model1DBatch
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
def report(name, model, per, kind, nlay=None, flags=0):
    B, _, L = model.shape; P = per.numel()
    plan = forward.BatchPlan(B, L, P)
    c, u, st = plan.run(model, per, kind=kind | flags, nlay=nlay)
    torch.cuda.synchronize()
    fb, nv, el = plan.counters()
    print(f"{name}: B={B} L={L} P={P} kind={kind}: exact-fallback stacks {fb} ({fb / B:.2e}), NEVILL-by-phase {nv} ({nv / (B * P):.2e} of the (stack, period) pairs), "
          f"ellipticities redone {el} ({el / (B * P):.2e}), solved {float((st == 0).float().mean()):.4f}", flush=True)
per20 = torch.from_numpy(synth.default_periods(20)).to(dev)
m = torch.from_numpy(synth.synth_models(65536, 10, seed=0)).to(dev)
report("forward L10 R", m, per20, 2); report("forward L10 L", m, per20, 1)
mb = Model1DBatch(bench.C5_SETTING, device=dev)
params = TorchProposer(mb.spec, dev, seed=1).reset(16384)
model, nlay = mb.to_model(params)
report("c5 thermal R", model.contiguous(), per20, 2, nlay); report("c5 thermal L", model.contiguous(), per20, 1, nlay)
from pysurfinv_amd.settings import MCMC_SETTING, MCMC_PERIODS
mg = Model1DBatch(MCMC_SETTING, device=dev)
pg = TorchProposer(mg.spec, dev, seed=3).reset(25600)
mod, nl = mg.to_model(pg)
perg = torch.as_tensor(np.asarray(MCMC_PERIODS, np.float32), device=dev)
report("grid prior draws R (phase only)", mod.contiguous(), perg, 2, nl, flags=_lib.PHASE_ONLY)
