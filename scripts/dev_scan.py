#!/usr/bin/env python3
"""Development: default (certified coarse-to-fine) scan against SURFDISP_EXACTSCAN - identical outputs? how much faster?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_cases
from pysurfinv_amd import _lib, forward, synth

L = _lib.lib()
cases = load_cases()
tot = dif = 0
for name in sorted(cases):
    d = cases[name]
    for t in (2, 4, 8):
        L.surfdisp_set_team(t)
        c0, u0, s0 = forward.forward_batch(d["model"], d["periods"], d["kind"], exact_scan=True)
        c1, u1, s1 = forward.forward_batch(d["model"], d["periods"], d["kind"])
        nd = int((c0 != c1).sum())
        tot += c0.size; dif += nd
        if nd:
            print(f"{name:24s} team {t}: {nd} of {c0.size} entries differ, max |dc| {np.abs(c0 - c1).max():.2e}, patterns {int(((c0 > 0) != (c1 > 0)).sum())}")
print(f"golden cases: {dif} of {tot} phase velocities differ between the two scans")
L.surfdisp_set_team(0)
# random rough stacks
rng = np.random.default_rng(1)
tot = dif = pat = 0
for it in range(int(os.environ.get("FS_CASES", "60"))):
    Ln = int(rng.integers(3, 40)); B = 2048; kind = int(rng.integers(1, 3))
    m = synth.synth_models(B, Ln, seed=int(rng.integers(1 << 30)), noise=float(rng.choice([0.02, 0.05, 0.1, 0.2])),
                           monotone=bool(rng.random() < 0.5), total_thickness=float(rng.choice([60., 120., 200.])))
    P = int(rng.integers(5, 30)); per = np.sort(rng.uniform(4.0, 120.0, P)).astype(np.float32)
    L.surfdisp_set_team(int(rng.choice([2, 4, 8])))
    c0, u0, s0 = forward.forward_batch(m, per, kind, exact_scan=True)
    c1, u1, s1 = forward.forward_batch(m, per, kind)
    tot += c0.size; dif += int((c0 != c1).sum()); pat += int(((c0 > 0) != (c1 > 0)).any(axis=1).sum())
print(f"random stacks: {dif} of {tot} phase velocities differ ({dif / tot:.2e}); stacks with a different zero pattern: {pat}")
L.surfdisp_set_team(0)
# timing
per = torch.from_numpy(synth.default_periods(20)).cuda()
for (B, Ln, kind) in ((65536, 10, 2), (65536, 10, 1), (8192, 64, 2)):
    model = torch.from_numpy(synth.synth_models(B, Ln, seed=0)).cuda()
    plan = forward.BatchPlan(B, Ln, 20)
    for t in (0, 2):
        L.surfdisp_set_team(t)
        for fs in (True, False):
            plan.run(model, per, kind=kind | 0x10, exact_scan=fs); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): plan.run(model, per, kind=kind | 0x10, exact_scan=fs)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(f"B={B} L={Ln} kind={kind} team={t} exact_scan={fs}: phase-only {dt*1e3:7.2f} ms")
L.surfdisp_set_team(0)
