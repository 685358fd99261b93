#!/usr/bin/env python3
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward, synth
import torch
per = torch.from_numpy(synth.default_periods(20)).cuda()
kind = int(os.environ.get("KIND", "2")); Ln = int(os.environ.get("L", "10"))
for B in [int(x) for x in os.environ.get("BS", "16384,65536,131072,262144").split(",")]:
    model = torch.from_numpy(synth.synth_models(B, Ln, seed=0)).cuda()
    plan = forward.BatchPlan(B, Ln, 20)
    for t in [int(x) for x in os.environ.get("TTEAMS", "1,2,4,8").split(",")]:
        _lib.lib().surfdisp_set_team(t)
        plan.run(model, per, kind=kind); torch.cuda.synchronize()
        ms = np.zeros(3)
        for _ in range(3):
            *_, m = plan.run_timed(model, per, kind=kind); ms += np.array(m)
        ms /= 3
        print(f"B={B} L={Ln} kind={kind} team={t}: phase {ms[1]:7.3f} ms group {ms[2]:6.3f} ms  -> phase-only {B/ms[1]/1e3:7.2f} Msolves/s", flush=True)
_lib.lib().surfdisp_set_team(0)
