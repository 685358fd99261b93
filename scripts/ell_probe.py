#!/usr/bin/env python3
"""Developer aid: ellipticity and group velocity of saved soak offenders (scripts/soak.py) from the HIP path beside the oracle's,
for the team size of the soak case.  usage: ell_probe.py file.npz index [index ...]   (SURFDISP_ELL_AMBIG is read by the library)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward
from oracle import cport
f = np.load(sys.argv[1])
L_ = cport.lib(); fp = ctypes.POINTER(ctypes.c_float)
L_.surfdisp_oracle_forward_dbg.restype = ctypes.c_int
L_.surfdisp_oracle_forward_dbg.argtypes = [ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp, fp, ctypes.c_int, fp, fp, fp]
for q in map(int, sys.argv[2:]):
    n = int(f["nlay"][q]); P = int(f["P"][q]); kind = int(f["kind"][q]); team = int(f["team"][q])
    m = np.ascontiguousarray(f["model"][q][:, :n]); per = np.ascontiguousarray(f["per"][q][:P])
    co = np.zeros(P, np.float32); uo = np.zeros(P, np.float32); ro = np.zeros(P, np.float32)
    rows = [np.ascontiguousarray(m[j]) for j in range(5)]
    L_.surfdisp_oracle_forward_dbg(n, kind, *[x.ctypes.data_as(fp) for x in rows], per.ctypes.data_as(fp), P,
                                   co.ctypes.data_as(fp), uo.ctypes.data_as(fp), ro.ctypes.data_as(fp))
    _lib.lib().surfdisp_set_team(team)
    plan = forward.BatchPlan(1, n, P)
    c, u, st, r = plan.run(torch.from_numpy(m[None]).cuda(), torch.from_numpy(per).cuda(), kind=kind, want_ratio=True)
    c, u, r = c.cpu().numpy()[0], u.cpu().numpy()[0], r.cpu().numpy()[0]
    with np.errstate(all="ignore"):
        k = int(np.nanargmax(np.abs(u / uo - 1)))
    print(f"#{q} team={team} L={n} k={k} T={per[k]:.3f}: c {c[k]:.7f} / {co[k]:.7f}   ratio {r[k]:.7f} / {ro[k]:.7f} ({abs(r[k]/ro[k]-1):.1e})   U {u[k]:.7f} / {uo[k]:.7f} ({abs(u[k]/uo[k]-1):.1e})", flush=True)
_lib.lib().surfdisp_set_team(0)
