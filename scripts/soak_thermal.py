#!/usr/bin/env python3
"""GPU soak: native thermal kernel (surfdisp_thermal_kernel + surfdisp_layers_kernel) against the torch
mirror on random draws from the prior boxes of the two static hybrid settings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from settings_therm import HYBRID_STATIC, HYBRID_STATIC_YAMA
from pysurfinv_amd.layers_batch import Model1DBatch
from pysurfinv_amd.brownian import TorchProposer

T_END = time.time() + float(os.environ.get("SOAK_SECONDS", "60"))
tot = 0; worst = 0.0; worst_grid = 0.0; nbad = 0
seed = 0
while time.time() < T_END:
    for name, setting in (("ritz", HYBRID_STATIC), ("yama", HYBRID_STATIC_YAMA)):
        mb = Model1DBatch(setting, device="cuda:0")
        seed += 1
        p = TorchProposer(mb.spec, "cuda:0", seed=seed).reset(8192)
        m_nat, _ = mb.to_model(p)
        m_ref, _ = mb.to_model_torch(p)
        d = ((m_nat - m_ref).abs() / m_ref.abs().clamp(min=1e-3)).max(dim=2).values.max(dim=1).values
        vs_t, qs_t = mb.layers[-1]["grid_last"]
        sc = mb._thermal_scratch[:, :vs_t.shape[1]]
        g = torch.maximum(((sc[:, :, 0] - vs_t).abs() / vs_t.abs()).max(dim=1).values,
                          ((sc[:, :, 1] - qs_t).abs() / qs_t.abs()).max(dim=1).values)
        tot += p.shape[0]; worst = max(worst, float(d.max())); worst_grid = max(worst_grid, float(g.max()))
        nbad += int((d > 2e-6).sum())
        if float(d.max()) > 2e-6:
            i = int(d.argmax()); print(name, "worst draw", p[i].tolist(), float(d[i]), float(g[i]), flush=True)
    print(f"  ... {tot} chains, worst fp32 stack difference {worst:.2e}, worst fp64 grid difference {worst_grid:.2e}, above 2e-6: {nbad}", flush=True)
print(f"thermal soak: {tot} chains, worst relative difference of the fp32 stacks {worst:.2e}, of the fp64 grid values {worst_grid:.2e}; chains above 2e-6: {nbad}")
