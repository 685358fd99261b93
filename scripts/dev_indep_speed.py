#!/usr/bin/env python3
"""Latency of small batches: faithful vs independent mode (root-search kernel, HIP events)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import forward, synth
per = torch.from_numpy(synth.default_periods(20)).cuda()
for L in (10, 96):
    for B in (1, 100, 1500, 8192, 65536 if L == 10 else 25600):
        model = torch.from_numpy(synth.synth_models(B, L, seed=0)).cuda()
        plan = forward.BatchPlan(B, L, 20)
        out = []
        for ind in (False, True):
            plan.run(model, per, kind=2, independent=ind); torch.cuda.synchronize()
            ms = np.zeros(3)
            for _ in range(3):
                *_, m = plan.run_timed(model, per, kind=2, independent=ind); ms += np.array(m)
            out.append(ms / 3)
        print(f"L={L:3d} B={B:6d}: phase faithful {out[0][1]:8.3f} ms | independent {out[1][1]:8.3f} ms   (group {out[0][2]:.3f})", flush=True)
