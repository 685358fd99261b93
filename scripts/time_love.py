#!/usr/bin/env python3
"""Development timing of the Love path: kernel durations (HIP events) of a 65 536 x L10 and a 16 384 x L64 batch,
c + U at 20 periods.  A/B: SURFDISP_LIB_PATH=.../libsurfdisp_var.so python scripts/time_love.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import forward, synth
per = torch.from_numpy(synth.default_periods(20)).cuda()
for B, L in ((65536, 10), (65536, 30), (16384, 64), (25600, 96)):
    m = torch.from_numpy(synth.synth_models(B, L, seed=1)).cuda()
    plan = forward.BatchPlan(B, L, 20)
    for _ in range(3):
        plan.run(m, per, kind=1)
    torch.cuda.synchronize()
    ts = np.array([plan.run_timed(m, per, kind=1)[-1] for _ in range(10)])
    print(f"Love {B} x L{L}: prep / phase / group ms = {np.round(ts.mean(0), 4)}", flush=True)
    for _ in range(2):
        plan.run(m, per, kind=1 | 0x80)
    torch.cuda.synchronize()
    ts = np.array([plan.run_timed(m, per, kind=1 | 0x80)[-1] for _ in range(10)])        # SURFDISP_EXACTSCAN: every grid point
    print(f"   point-by-point scan:     prep / phase / group ms = {np.round(ts.mean(0), 4)}", flush=True)
