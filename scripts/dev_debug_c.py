#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cases
from pysurfinv_amd import _lib, forward
np.set_printoptions(linewidth=250, precision=7)
cases = load_cases()
for name in sys.argv[1:]:
    d = cases[name]
    for wt in os.environ.get("WTOLS", "3.2e-4").split(","):
        os.environ["SURFDISP_WTOL"] = wt
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"])
        ec = np.abs(c / d["c"] - 1); eu = np.abs(u / d["u"] - 1)
        b = np.unravel_index(np.nanargmax(ec), ec.shape)[0]
        print(name, "wtol", wt, "worst stack", b, "max errC", np.nanmax(ec), "max errU", np.nanmax(eu))
        print(" errC", ec[b]); print(" errU", eu[b]); print(" c   ", c[b]); print(" cref", d["c"][b])
        print(" per-period max errC over stacks", np.nanmax(ec, axis=0))
