#!/usr/bin/env python3
"""Development: root-search time with / without the ellipticity snapshot slot, by team size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import _lib, forward, synth
L = _lib.lib()
per = torch.from_numpy(synth.default_periods(20)).cuda()
for (B, Ln) in ((65536, 10), (32768, 20), (16384, 32), (8192, 64), (25600, 96)):
    model = torch.from_numpy(synth.synth_models(B, Ln, seed=0)).cuda()
    plan = forward.BatchPlan(B, Ln, 20)
    for cap in os.environ.get("CAPS", "65536").split(","):
        os.environ["SURFDISP_OVERLAP_MAX"] = cap
        for team in [int(x) for x in os.environ.get("TEAMS", "0,4,8,16,32").split(",")]:
            if L.surfdisp_set_team(team) != 0: continue
            try:
                plan.run(model, per, kind=2 | 0x10); torch.cuda.synchronize()
            except Exception as e:
                print(f"B={B} L={Ln} cap={cap} team={team}: {e}"); continue
            t0 = time.perf_counter()
            for _ in range(3): plan.run(model, per, kind=2 | 0x10)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            print(f"B={B} L={Ln} overlap_max={cap:>8s} team={team:2d} (auto -> {L.surfdisp_get_team(B, Ln)}): {dt*1e3:7.2f} ms", flush=True)
L.surfdisp_set_team(0)
