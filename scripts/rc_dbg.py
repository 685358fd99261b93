import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from pysurfinv_amd import _lib, forward
f = np.load(sys.argv[1]); i = int(sys.argv[2])
n = int(f["nlay"][i]); P = int(f["P"][i]); team = int(f["team"][i])
m = torch.from_numpy(np.ascontiguousarray(f["model"][i][:, :n])[None].copy()).cuda()
per = torch.from_numpy(f["per"][i][:P].copy()).cuda()
plan = forward.BatchPlan(1, n, P)
_lib.lib().surfdisp_set_team(team)
c, u, st = plan.run(m, per, kind=2 | _lib.PHASE_ONLY)
torch.cuda.synchronize()
print(c.cpu().numpy()[0][:8])
