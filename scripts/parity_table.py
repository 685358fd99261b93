#!/usr/bin/env python3
"""CPU: per golden case, zero-pattern mismatches and the c / U error distribution of a dump made on the GPU box by
scripts/parity_dump.py (gpurun_out/parity_dump.npz)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cases, load_families

z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_dump.npz"))
cases = dict(load_cases())
cases.update({'fam_' + k: v for k, v in load_families().items()})
tags = sorted({k.split("/")[1] for k in z.files})
print(f"{'case':22s} {'tag':5s} {'zero-pattern stacks':>20s} {'c max':>9s} {'U q50':>9s} {'U q99':>9s} {'U max':>9s}  n(U>5e-5) n(U>1e-4)")
for name, d in sorted(cases.items()):
    for tag in tags:
        c, u = z[f"{name}/{tag}/c"], z[f"{name}/{tag}/u"]
        rows = ((c > 0) == (d["c"] > 0)).all(axis=1)
        ok = (d["c"] != 0) & rows[:, None]
        if not ok.any():
            print(f"{name:22s} {tag:5s} {int((~rows).sum()):>8d} of {len(rows):<8d} (nothing solved)")
            continue
        ec = np.abs(c[ok].astype(np.float64) / d["c"][ok] - 1)
        oku = ok & (d["u"] != 0) & np.isfinite(d["u"]) & (np.abs(d["u"]) > 1e-3)
        eu = np.abs(u[oku].astype(np.float64) / d["u"][oku] - 1)
        eu = np.where(np.isfinite(eu), eu, 1.0)
        nanpat = int(((~np.isfinite(u)) != (~np.isfinite(d["u"])))[ok].sum())
        if tag in ("t0",) or eu.max() > 5e-5 or (~rows).any():
            print(f"{name:22s} {tag:5s} {int((~rows).sum()):>8d} of {len(rows):<8d} {ec.max():9.2e} {np.median(eu):9.2e} "
                  f"{np.quantile(eu, 0.99):9.2e} {eu.max():9.2e}  {int((eu > 5e-5).sum()):6d} {int((eu > 1e-4).sum()):6d}  st {np.bincount(z[f'{name}/{tag}/st'], minlength=9)[[0,1,2,4,8]]} nanpat {nanpat}")
