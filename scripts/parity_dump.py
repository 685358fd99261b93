#!/usr/bin/env python3
"""GPU box: run every golden case (tests/golden/ref_cases.npz) and every soak-family fixture
(tests/golden/ref_families.npz) through the HIP path - default team and every team size, default scan and the
opt-in fast scan - and dump the raw outputs to gpurun_out/parity_dump.npz for offline analysis
(scripts/parity_table.py, CPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cases, load_families
from pysurfinv_amd import _lib, forward

cases = dict(load_cases())
cases.update({"fam_" + k: v for k, v in load_families().items()})
out = {}
for name, d in sorted(cases.items()):
    nl = d.get("nlay")
    for team in (0, 1, 2, 4, 8, 16, 32, 64):
        _lib.lib().surfdisp_set_team(team)
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"], nlay=nl)
        out[f"{name}/t{team}/c"] = c; out[f"{name}/t{team}/u"] = u; out[f"{name}/t{team}/st"] = st
    for team in (2, 4, 8):
        _lib.lib().surfdisp_set_team(team)
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"], nlay=nl, fast_scan=True)
        out[f"{name}/fast{team}/c"] = c; out[f"{name}/fast{team}/u"] = u; out[f"{name}/fast{team}/st"] = st
    _lib.lib().surfdisp_set_team(0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "parity_dump.npz"), **out)
print("wrote gpurun_out/parity_dump.npz with", len(out), "arrays")
