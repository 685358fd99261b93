"""GPU, developer build -DSD_COUNT_KINK (SURFDISP_LIB_PATH): how many refine passes the kink test of the root search adds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from pysurfinv_amd import forward, synth
per = torch.from_numpy(synth.default_periods(20)).cuda()
for kind in (1, 2):
    for B, L in ((65536, 10), (16384, 64)):
        m = torch.from_numpy(synth.synth_models(B, L, seed=1)).cuda()
        plan = forward.BatchPlan(B, L, 20)
        plan.run(m, per, kind=kind | 0x10)
        torch.cuda.synchronize()
        fb, nv, el = plan.counters()
        print(f"kind {kind} B {B} L {L}: refine passes added by the kink test {el}; NEVILL by phase {nv}; units {B*20}", flush=True)
