"""Developer aid (GPU): one saved soak offender, the first periods of c, U and ellipticity for every team size in default and
SURFDISP_STRICT mode beside the oracle.  usage: u_probe2.py soak_offenders_X.npz index"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from pysurfinv_amd import _lib, forward
from oracle import cport, refso
f = np.load(sys.argv[1]); i = int(sys.argv[2])
np.set_printoptions(linewidth=220, precision=7, suppress=True)
n = int(f["nlay"][i]); P = int(f["P"][i]); kind = int(f["kind"][i])
m = np.ascontiguousarray(f["model"][i][:, :n])[None].copy(); per = f["per"][i][:P].copy()
co, uo, so = cport.forward_batch(m, per, kind)
print("oracle c", co[0][:4], "U", uo[0][:4])
try:
    r = refso.fast_surf(n, kind, m[0][0], m[0][1], m[0][2], m[0][3], m[0][4], per, P)
    print("reference c", r[2][:4], "U", r[0][:4], "ratio", refso.last_ratio()[:4])
except Exception as e:
    print("refso", e)
plan = forward.BatchPlan(1, n, P)
mt, pt = torch.from_numpy(m).cuda(), torch.from_numpy(per).cuda()
for tm in (1, 2, 4, 8, 16, 64):
    _lib.lib().surfdisp_set_team(tm)
    for name, fl in (("default", 0), ("strict", _lib.STRICT)):
        c, u, st, r = plan.run(mt, pt, kind=kind | fl, want_ratio=True)
        torch.cuda.synchronize()
        print(f"team {tm:2d} {name:7s} c {c.cpu().numpy()[0][:4]} U {u.cpu().numpy()[0][:4]} ratio {r.cpu().numpy()[0][:4]}")
