#!/usr/bin/env python3
"""Developer aid: re-run the offending stacks a soak saved (scripts/soak.py -> soak_offenders_*.npz) one by one through the
HIP path in several modes (recorded team size, other team sizes, every grid point, SURFDISP_STRICT) beside the CPU oracle, to
see WHICH decision differs.  usage: offender_probe.py file.npz cat [max] [first]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward
from oracle import cport

f = np.load(sys.argv[1]); cat = int(sys.argv[2]); nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 30
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
idx = np.nonzero(f["cat"] == cat)[0][first:]
step = max(1, len(idx) // nmax)
L = _lib.lib()
np.set_printoptions(linewidth=220, precision=6)
agree = {}
for i in idx[::step][:nmax]:
    n = int(f["nlay"][i]); P = int(f["P"][i]); kind = int(f["kind"][i]); team = int(f["team"][i])
    m = np.ascontiguousarray(f["model"][i][:, :n])[None]; per = f["per"][i][:P].copy()
    co, uo, so = cport.forward_batch(m, per, kind)
    c_rec = f["c"][i][:P]
    with np.errstate(all="ignore"):
        e = np.abs(c_rec / co[0] - 1) if cat == 1 else np.abs(f["u"][i][:P] / uo[0] - 1)
    k = int(np.nanargmax(np.nan_to_num(e, nan=9.0)))
    out = {}
    for name, tm, kw in (("rec", team, {}), ("t1", 1, {}), ("t2", 2, {}), ("t4", 4, {}), ("t8", 8, {}), ("t16", 16, {}), ("t64", 64, {}),
                         ("strict", team, dict(strict=True))):
        L.surfdisp_set_team(tm)
        c, u, st = forward.forward_batch(m, per, kind, **kw)
        out[name] = (c[0], u[0], int(st[0]))
    os.environ["X"] = "1"
    L.surfdisp_set_team(0)
    line = f"i={i} kind={kind} team={team} L={n} P={P} k={k} T={per[k]:.4f} oracle c={co[0][k]:.6f} u={uo[0][k]:.6f} | recorded c={c_rec[k]:.6f} u={f['u'][i][k]:.6f} |"
    for name, (c, u, st) in out.items():
        okc = abs(c[k] / co[0][k] - 1) < 2e-5 if co[0][k] else c[k] == 0
        with np.errstate(all="ignore"):
            oku = abs(u[k] / uo[0][k] - 1) < 1e-4
        agree[name] = agree.get(name, 0) + int(okc and (cat == 1 or oku))
        line += f" {name}:{c[k]:.6f}/{u[k]:.6f}{'' if okc else '*'}{'' if oku else '!'}"
    print(line, flush=True)
print("agreements with the oracle at the worst period (of", len(idx[::step][:nmax]), "):", agree)
