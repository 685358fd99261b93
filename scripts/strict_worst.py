#!/usr/bin/env python3
"""Developer aid: default vs SURFDISP_STRICT on the 16 384 x L64 test batch - the worst phase-velocity entries, beside the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import synth, forward
from oracle import cport
B, L = 16384, 64
per = synth.default_periods(20)
mh = synth.synth_models(B, L, seed=1)
m = torch.from_numpy(mh).cuda(); pt = torch.from_numpy(per).cuda()
plan = forward.BatchPlan(B, L, 20)
plan.run(m, pt, kind=2, strict=True); cs = plan.c.cpu().numpy().copy()
plan.run(m, pt, kind=2); cd = plan.c.cpu().numpy().copy()
print("counters", plan.counters())
e = np.abs(cd.astype(np.float64) / cs - 1)
idx = np.argsort(e.ravel())[::-1][:8]
for q in idx:
    b, k = divmod(int(q), 20)
    co = cport.forward_batch(mh[b:b + 1], per, 2)[0][0]
    print(f"stack {b} k={k} T={per[k]:.2f}: default {cd[b, k]:.7f} strict {cs[b, k]:.7f} oracle {co[k]:.7f}  |d-s| {e[b, k]:.1e} |d-o| {abs(cd[b,k]/co[k]-1):.1e} |s-o| {abs(cs[b,k]/co[k]-1):.1e}")
print("quantiles of |d/s-1|:", np.quantile(e, [0.5, 0.99, 0.9999, 1.0]))
