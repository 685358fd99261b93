#!/usr/bin/env python3
"""Developer probe (GPU box): parity table per golden case and team size + quick timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cases, relerr
from pysurfinv_amd import _lib, forward, synth
import torch

L = _lib.lib()
cases = load_cases()
teams = [int(x) for x in os.environ.get("TEAMS", "0,1,4,64").split(",")]
print(f"{'case':24s} " + " ".join(f"{'G=' + str(t):>24s}" for t in teams))
for name in sorted(cases):
    d = cases[name]
    row = []
    for t in teams:
        L.surfdisp_set_team(t)
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"])
        rows = ((c > 0) == (d["c"] > 0)).all(axis=1)
        row.append(f"{relerr(c[rows], d['c'][rows]):8.1e} {relerr(u[rows], d['u'][rows]):8.1e} {int((~rows).sum()):3d}")
    print(f"{name:24s} " + " ".join(f"{r:>24s}" for r in row), flush=True)
L.surfdisp_set_team(0)

if os.environ.get("TIMING", "1") == "1":
    per = torch.from_numpy(synth.default_periods(20)).cuda()
    for (B, Ln, kind) in ((65536, 10, 2), (65536, 10, 1), (8192, 64, 2)):
        model = torch.from_numpy(synth.synth_models(B, Ln, seed=0)).cuda()
        plan = forward.BatchPlan(B, Ln, 20)
        for t in [int(x) for x in os.environ.get("TTEAMS", "1,2,4,8,16,64").split(",")]:
            if _lib.lib().surfdisp_set_team(t) != 0: continue
            try:
                plan.run(model, per, kind=kind); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3): plan.run(model, per, kind=kind)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
                print(f"B={B} L={Ln} kind={kind} team={t}: {dt*1e3:8.2f} ms  {B/dt/1e6:7.3f} Msolves/s", flush=True)
            except Exception as e:
                print(f"B={B} L={Ln} kind={kind} team={t}: FAILED {e}", flush=True)
    L.surfdisp_set_team(0)
