#!/usr/bin/env python3
"""Development sweep: root-search time for lanes per stack 2..64 against the library's own choice (team 0), over batch
shapes (stacks, layers, wave type, phase-only or c+U).  One batch in flight."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import forward, synth, _lib

per = torch.from_numpy(np.linspace(8, 80, 19).astype(np.float32)).cuda()
shapes = [(65536, 10, 2, False), (65536, 10, 1, False), (25600, 96, 2, True), (25600, 48, 2, True), (16384, 64, 2, False),
          (16384, 64, 1, False), (8192, 30, 2, False), (4096, 20, 2, True), (1024, 40, 2, True), (100, 96, 2, True)]
if os.environ.get('SWEEP_SHAPES'):
    shapes = [tuple(int(x) for x in t.split(',')[:3]) + (t.split(',')[3] == '1',) for t in os.environ['SWEEP_SHAPES'].split()]
for B, L, kind, ponly in shapes:
    m = torch.from_numpy(synth.synth_models(B, L, seed=1, total_thickness=220.0)).cuda()
    plan = forward.BatchPlan(B, L, 19)
    k = kind | (0x10 if ponly else 0)
    row = []
    for team in (0, 2, 4, 8, 16, 32, 64, 0):
        _lib.lib().surfdisp_set_team(team)
        for _ in range(2):
            plan.run(m, per, kind=k)
        torch.cuda.synchronize()
        ts = np.array([plan.run_timed(m, per, kind=k)[-1] for _ in range(4)])
        row.append((team, ts.mean(0)[1]))
    best = min(row[1:-1], key=lambda t: t[1])
    auto = min(row[0][1], row[-1][1])                      # measured first and last: the first config of a shape also pays the clock ramp
    print(f"B={B:6d} L={L:3d} kind={kind} {'phase-only' if ponly else 'c+U       '}: auto {row[0][1]:.3f} / {row[-1][1]:.3f} ms | " +
          "  ".join(f"{t}:{v:.3f}" for t, v in row[1:-1]) + f" | best {best[0]} ({auto / best[1]:.2f} x auto)", flush=True)
_lib.lib().surfdisp_set_team(0)
