#!/usr/bin/env python3
"""GPU box: the entries of tests/golden/u_exceptions_families.json (listed + unexplained), one by one: c, ellipticity and U of
the default root search and of SURFDISP_STRICT against the reference fixture and the CPU oracle (its ellipticity, and its U
evaluated AT the HIP path's c)."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_families
from oracle import cport
from pysurfinv_amd import _lib, forward
fam = load_families()
J = json.load(open(os.path.join(ROOT, "tests", "golden", "u_exceptions_families.json")))
O = cport.lib()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
for e in J.get("unexplained", []) + J["entries"]:
    d = fam[e["family"]]; b, k = e["stack"], e["period_index"]
    n = int(d["nlay"][b]); m = np.ascontiguousarray(d["model"][b:b + 1, :, :n], np.float32)
    per = np.ascontiguousarray(d["periods"], np.float32); P = len(per)
    co = np.zeros(P, np.float32); uo = np.zeros(P, np.float32); ro = np.zeros(P, np.float32)
    O.surfdisp_oracle_forward_dbg(n, 2, fp(m[0, 0]), fp(m[0, 1]), fp(m[0, 2]), fp(m[0, 3]), fp(m[0, 4]), fp(per), P, fp(co), fp(uo), fp(ro))
    print(f"{e['family']} stack {b} period {k} T={per[k]:.3f}: ref c {d['c'][b, k]:.7f} U {d['u'][b, k]:.7f} | oracle ratio {ro[k]:.6e}")
    dm, dp = torch.from_numpy(m).cuda(), torch.from_numpy(per).cuda()
    plan = forward.BatchPlan(1, n, P)
    for team in (0, 4, 16):
        _lib.lib().surfdisp_set_team(team)
        for mode, kw in (("default", {}), ("strict ", {"strict": True})):
            c, u, st, r = plan.run(dm, dp, kind=2, want_ratio=True, **kw)
            torch.cuda.synchronize()
            c, u, r = c.cpu().numpy()[0], u.cpu().numpy()[0], r.cpu().numpy()[0]
            _, ua = cport.group_at(m, per, 2, c[None, :])
            print(f"   team {team:2d} {mode}: c {c[k]:.7f} (rel {c[k] / d['c'][b, k] - 1:+.2e})  ratio {r[k]:.6e} (rel {r[k] / ro[k] - 1:+.2e})  "
                  f"U {u[k]:.7f} (rel to ref {u[k] / d['u'][b, k] - 1:+.2e}; to the oracle's U at this c {u[k] / ua[0, k] - 1:+.2e})")
_lib.lib().surfdisp_set_team(0)
