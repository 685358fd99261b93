#!/usr/bin/env python3
"""GPU box: run every golden case through the HIP path (default team and every team size) and dump the raw
outputs to gpurun_out/parity_dump.npz for offline analysis (scripts/parity_table.py, CPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cases
from pysurfinv_amd import _lib, forward

cases = load_cases()
out = {}
for name, d in sorted(cases.items()):
    for team in (0, 1, 2, 4, 8, 16, 32, 64):
        _lib.lib().surfdisp_set_team(team)
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"])
        out[f"{name}/t{team}/c"] = c; out[f"{name}/t{team}/u"] = u; out[f"{name}/t{team}/st"] = st
    _lib.lib().surfdisp_set_team(0)
    c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"], fast_scan=True)
    out[f"{name}/fast/c"] = c; out[f"{name}/fast/u"] = u; out[f"{name}/fast/st"] = st
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "parity_dump.npz"), **out)
print("wrote gpurun_out/parity_dump.npz with", len(out), "arrays")
