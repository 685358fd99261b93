#!/usr/bin/env python3
"""Developer aid: one (family, stack, period) entry of tests/golden/ref_families.npz through the HIP path for several team
sizes beside the reference's value.  usage: entry_probe.py family stack period_index"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_families
from pysurfinv_amd import _lib, forward
fam, i, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
d = load_families()[fam]
m = d["model"][i:i + 1]; nl = d["nlay"][i:i + 1]
out = []
for team in (1, 2, 4, 8, 16, 64):
    _lib.lib().surfdisp_set_team(team)
    c, u, st = forward.forward_batch(m, d["periods"], d["kind"], nlay=nl)
    out.append(f"t{team}: c {c[0][k]:.7f} ({abs(c[0][k] / d['c'][i][k] - 1):.1e}) U {u[0][k]:.6f} ({abs(u[0][k] / d['u'][i][k] - 1):.1e})")
_lib.lib().surfdisp_set_team(0)
print(f"{fam}[{i}] k={k} T={d['periods'][k]:.3f} ref c {d['c'][i][k]:.7f} U {d['u'][i][k]:.6f} | " + " | ".join(out))
