#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cases, relerr
from pysurfinv_amd import forward
cases = load_cases()
for name in sorted(cases):
    d = cases[name]
    c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"], independent=True)
    rows = ((c > 0) == (d["c"] > 0)).all(axis=1)
    ok = d["c"][rows] != 0
    ec = np.abs(c[rows][ok] / d["c"][rows][ok] - 1) if ok.any() else np.zeros(1)
    eu = np.abs(u[rows][ok] / d["u"][rows][ok] - 1) if ok.any() else np.zeros(1)
    print(f"{name:22s} rows differ {int((~rows).sum()):3d}/{len(rows):3d}  c max {ec.max():.1e} 99% {np.quantile(ec,0.99):.1e}  U max {eu.max():.1e} 99% {np.quantile(eu,0.99):.1e}", flush=True)
