#!/usr/bin/env python3
"""Development: root-search + group time by team size for mid-size batches of shallow stacks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import _lib, forward, synth
L = _lib.lib()
per = torch.from_numpy(synth.default_periods(20)).cuda()
for (B, Ln) in ((4096, 10), (8192, 10), (16384, 10), (32768, 10), (16384, 20), (8192, 32)):
    model = torch.from_numpy(synth.synth_models(B, Ln, seed=0)).cuda()
    plan = forward.BatchPlan(B, Ln, 20)
    row = []
    for team in (0, 2, 4, 8, 16, 32, 64):
        if L.surfdisp_set_team(team) != 0: continue
        for _ in range(2): plan.run(model, per, kind=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): plan.run(model, per, kind=2)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        row.append(f"G={team if team else 'auto(' + str(L.surfdisp_get_team(B, Ln)) + ')'}: {dt*1e3:.2f}")
    print(f"B={B} L={Ln} c+U ms: " + "  ".join(row), flush=True)
L.surfdisp_set_team(0)
