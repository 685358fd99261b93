"""Developer aid (GPU), with a library built -DSD_DEBUG_TRIALS=<period> -DSD_DEBUG_CMIN=<km/s> (SURFDISP_LIB_PATH): prints every
trial of that period of ONE saved soak offender above that velocity, and the number of layers each period rebuilds.
usage: dbg_trials.py soak_offenders_X.npz index team"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from pysurfinv_amd import _lib, forward
f = np.load(sys.argv[1]); i = int(sys.argv[2]); tm = int(sys.argv[3])
n = int(f["nlay"][i]); P = int(f["P"][i]); kind = int(f["kind"][i])
m = torch.from_numpy(np.ascontiguousarray(f["model"][i][:, :n])[None].copy()).cuda()
per = torch.from_numpy(f["per"][i][:P].copy()).cuda()
plan = forward.BatchPlan(1, n, P)
_lib.lib().surfdisp_set_team(tm)
c, u, st = plan.run(m, per, kind=kind | 0x80)
torch.cuda.synchronize()
print(c.cpu().numpy())
