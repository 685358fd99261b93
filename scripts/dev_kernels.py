#!/usr/bin/env python3
"""Development: analytic partials (surfdisp_forward_kernels_device) against central finite differences
of the HIP solver itself (1 % perturbations) and of the CPU oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import forward, synth
from oracle import cport

per_np = np.arange(10.0, 101.0, 10.0).astype(np.float32)
per = torch.from_numpy(per_np).cuda()
G = np.load(os.path.join(ROOT, "tests", "golden", "test1_eus.npz"))
cases = {"synth_L12": synth.synth_models(2, 12, seed=3)[:1], "eus_L68": G["model"].astype(np.float32)}
wm = synth.synth_models(1, 9, seed=5); wm[0, 1, 0] = 0.0; wm[0, 0, 0] = 1.5; wm[0, 2, 0] = 1.03; wm[0, 3, 0] = 3.0
cases["water_L9"] = wm
EPS = 0.01
for name, m in cases.items():
    L = m.shape[2]
    for kind in (2, 1):
        plan = forward.BatchPlan(1, L, per_np.size)
        c, u, st, kb, ka, kr = plan.run_kernels(torch.from_numpy(m).cuda(), per, kind=kind)
        torch.cuda.synchronize()
        kb, kr = kb[0].cpu().numpy(), kr[0].cpu().numpy()
        ka = ka[0].cpu().numpy() if ka is not None else None
        for row, an, label in ((1, kb, "Vs"), (0, ka, "Vp"), (2, kr, "rho")):
            if an is None:
                continue
            big = np.repeat(m, 2 * L, axis=0)
            for i in range(L):
                big[i, row, i] *= (1 - EPS); big[L + i, row, i] *= (1 + EPS)
            co, _, so = cport.forward_batch(big, per_np, kind, nthreads=8)
            fd = ((co[L:].astype(np.float64) - co[:L]) / (2 * EPS * m[0, row][:, None])).T      # [P, L]
            fd[:, m[0, row] == 0] = 0
            scale = np.abs(fd).max(axis=1, keepdims=True)
            err = np.abs(an - fd) / scale
            print(f"{name:10s} kind={kind} d/d{label:3s}: max |analytic - FD| / max|FD| per period = "
                  f"{err.max():.2e}  (worst layer {np.unravel_index(err.argmax(), err.shape)}), "
                  f"sum check {np.abs(an.sum(1) - fd.sum(1)).max() / np.abs(fd.sum(1)).max():.1e}")
