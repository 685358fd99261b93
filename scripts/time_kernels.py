#!/usr/bin/env python3
"""GPU: time surfdisp_forward_kernels_device (forward + analytic partials) against the plain forward solve for the c5 shape
(16 384 x L64 x P20 thermal-like stacks), Rayleigh and Love.  Whole calls between torch events on the launch stream
(prep + root search + ellipticity + group-velocity kernel [+ transposition]); the group-velocity kernel's own share is the
difference of the two columns' last kernels - use rocprofv3 --kernel-trace --stats on this script for per-kernel times."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import forward, synth, _lib

B, L, P = int(os.environ.get("TK_B", 16384)), int(os.environ.get("TK_L", 64)), 20
m = torch.from_numpy(synth.synth_models(B, L, seed=1, noise=0.02, total_thickness=300.0)).cuda()
per = torch.from_numpy(synth.default_periods(P)).cuda()
plan = forward.BatchPlan(B, L, P)
def timed(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = [os.environ.get("SURFDISP_LIB_PATH", "default").split("/")[-1]]
for kind, w in ((2, "R"), (1, "L")):
    t_plain = timed(lambda: plan.run(m, per, kind=kind))
    t_kern = timed(lambda: plan.run_kernels(m, per, kind=kind))
    c, u, st, kb, ka, kr = plan.run_kernels(m, per, kind=kind)
    torch.cuda.synchronize()
    chk = float(kb.double().abs().sum().item())
    out.append(f"{w}: plain {t_plain:.3f} ms  kernels {t_kern:.3f} ms  (+{t_kern - t_plain:.3f})  sum|dcdb| {chk:.9e}  solved {int((st == 0).sum().item())}")
print("  ".join(out), flush=True)
