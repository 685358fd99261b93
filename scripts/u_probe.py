"""Developer aid (GPU): group velocity and ellipticity of saved soak offenders at their worst period for several team sizes, under
whatever SURFDISP_* environment is set (e.g. SURFDISP_ELL_AMBIG=-1: every ellipticity with the reference arithmetic).
usage: u_probe.py soak_offenders_X.npz index ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from pysurfinv_amd import _lib, forward
from oracle import cport
f = np.load(sys.argv[1])
np.set_printoptions(linewidth=220, precision=7)
for i in [int(x) for x in sys.argv[2:]]:
    n = int(f["nlay"][i]); P = int(f["P"][i]); kind = int(f["kind"][i]); team = int(f["team"][i])
    m = np.ascontiguousarray(f["model"][i][:, :n])[None].copy(); per = f["per"][i][:P].copy()
    co, uo, so = cport.forward_batch(m, per, kind)
    plan = forward.BatchPlan(1, n, P)
    mt, pt = torch.from_numpy(m).cuda(), torch.from_numpy(per).cuda()
    for tm in (team, 1, 4, 16):
        _lib.lib().surfdisp_set_team(tm)
        c, u, st, r = plan.run(mt, pt, kind=kind, want_ratio=True)
        torch.cuda.synchronize()
        u = u.cpu().numpy()[0]; c = c.cpu().numpy()[0]; r = r.cpu().numpy()[0]
        with np.errstate(all="ignore"):
            eu = np.abs(u / uo[0] - 1); k = int(np.nanargmax(np.nan_to_num(eu)))
        print(f"i={i} env ELL_AMBIG={os.environ.get('SURFDISP_ELL_AMBIG')} team {tm}: worst period {k} T={per[k]:.4f} c {c[k]:.7f} (oracle {co[0][k]:.7f}) U {u[k]:.7f} (oracle {uo[0][k]:.7f}) err {eu[k]:.2e} ratio {r[k]:.6f} counters {plan.counters()}")
    print("   model vs", m[0][1], "h", m[0][3])
