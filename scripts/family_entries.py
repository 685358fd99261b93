#!/usr/bin/env python3
"""GPU box: the soak-family fixtures (tests/golden/ref_families.npz) through the default root search and through
SURFDISP_STRICT, every team size the parity test uses -> gpurun_out/family_entries.npz (c, u per family / team / mode).
Input of tests/golden/make_golden_spread_families.py, which lists the entries the 1e-4 bar cannot hold."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_families
from pysurfinv_amd import _lib, forward
out = {}
for fam, d in sorted(load_families().items()):
    if fam.startswith("wild"):
        continue
    for team in (0, 1, 4, 16):
        _lib.lib().surfdisp_set_team(team)
        for mode, kw in (("default", {}), ("strict", {"strict": True})):
            c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"], nlay=d["nlay"], **kw)
            out[f"{fam}/{team}/{mode}/c"], out[f"{fam}/{team}/{mode}/u"] = c, u
    _lib.lib().surfdisp_set_team(0)
    c, u = out[f"{fam}/0/default/c"], out[f"{fam}/0/default/u"]
    ok = np.isfinite(d["u"]) & (np.abs(d["u"]) > 1e-3) & (c > 0) & (d["c"] > 0)
    e = np.abs(u[ok].astype(np.float64) / d["u"][ok] - 1)
    print(fam, "entries", int(ok.sum()), "U err > 1e-4:", int((~(e <= 1e-4)).sum()), "max", float(np.nanmax(e)))
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "family_entries.npz"), **out)
