#!/usr/bin/env python3
"""Developer aid (GPU): ONE saved soak offender (scripts/soak.py -> soak_offenders_*.npz) through the device entry for several team
sizes and scan flags, under whatever SURFDISP_* environment the caller set: one_offender.py file.npz index"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward
f = np.load(sys.argv[1]); i = int(sys.argv[2])
n = int(f["nlay"][i]); P = int(f["P"][i]); kind = int(f["kind"][i])
m = torch.from_numpy(np.ascontiguousarray(f["model"][i][:, :n])[None].copy()).cuda()
per = torch.from_numpy(f["per"][i][:P].copy()).cuda()
co = f["co"][i][:P]
np.set_printoptions(linewidth=250, precision=6)
print("env", {k: v for k, v in os.environ.items() if k.startswith("SURFDISP_") and k != "SURFDISP_LIB_PATH"}, "oracle", co)
plan = forward.BatchPlan(1, n, P)
for tm in (1, 2, 4, 8, 16, 64):
    _lib.lib().surfdisp_set_team(tm)
    for name, flags in (("default", 0), ("every grid point", 0x80), ("independent", 0x20)):
        c, u, st = plan.run(m, per, kind=kind | flags)
        torch.cuda.synchronize()
        c = c.cpu().numpy()[0]
        with np.errstate(all="ignore"):
            bad = np.nonzero(np.abs(c / co - 1) > 2e-5)[0]
        print(f"  team {tm:2d} {name:17s} counters {plan.counters()}  periods off by > 2e-5: {list(bad)}  c there {c[bad]}")
