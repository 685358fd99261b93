#!/usr/bin/env python3
"""Re-run saved soak failures (gpurun_out/soak_fail_*.npz) under different refine settings."""
import os, sys, glob
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward
np.set_printoptions(linewidth=200, precision=7)
for f in sorted(glob.glob(os.path.join(os.environ.get("SOAK_DIR", os.path.join(ROOT, "gpurun_out")), "soak_fail_*.npz"))):
    d = np.load(f); kind = int(d["kind"])
    out = []
    for wtol, atol, team in (("1.2e-3", "1e-6", int(d["team"])), ("1e-7", "1e-6", int(d["team"])), ("1e-7", "1e-6", 64), ("1e-7", "1e-6", 1)):
        os.environ["SURFDISP_WTOL"] = wtol; os.environ["SURFDISP_ATOL"] = atol
        _lib.lib().surfdisp_set_team(team)
        c, u, st = forward.forward_batch(d["model"], d["per"], kind, nlay=d["nlay"])
        ok = d["co"] != 0
        ec = np.abs(c[ok] / d["co"][ok] - 1)
        out.append(f"wtol={wtol} team={team}: c max {ec.max():.1e}")
    i = np.unravel_index(np.argmax(np.where(ok, np.abs(c / np.where(ok, d['co'], 1) - 1), 0)), c.shape)
    print(os.path.basename(f), f"kind={kind} L={d['model'].shape[2]}", " | ".join(out), f" worst (stack {i[0]}, T={d['per'][i[1]]:.2f}) c={c[i]:.6f} ref={d['co'][i]:.6f}")
_lib.lib().surfdisp_set_team(0)
