#!/usr/bin/env python3
"""BASELINE configs[4] kernel benchmark: joint R+L c+U on 64-layer stacks, one GPU."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import forward, synth
per = torch.from_numpy(synth.default_periods(20)).cuda()
for B in (8192, 32768):
    model = torch.from_numpy(synth.synth_models(B, 64, seed=0)).cuda()
    plan = forward.JointPlan(B, 64, 20)
    plan.run(model, per); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): plan.run(model, per)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"joint R+L c+U, L=64, B={B}: {dt*1e3:.2f} ms per batch = {B/dt/1e6:.3f} M stacks/s ({2*B/dt/1e6:.3f} M solves/s; 1600 B/stack algorithmic -> {1600*B/dt/1e9:.2f} GB/s)")
