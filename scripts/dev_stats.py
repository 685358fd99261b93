#!/usr/bin/env python3
"""Development: team passes per period by state.  Needs an instrumented build of the library
(atomic counters per state in the root-search kernel + an accessor surfdisp_stats) built by
`make -C pysurfinv_amd/csrc stats` into pysurfinv_amd/lib/libsurfdisp_stats.so; not part of the product build.  r01g, bench workload:
  team 2: scan 19.2 passes, refine 1.28, ellipticity 1.00 per period (43 evaluations; reference 52)
  team 4: scan 10.4, refine 1.07, ellipticity 0.05 (rides in the next scan pass)   (48 evaluations)
  default (certified coarse-to-fine) scan, r01j: team 2 scan 9.1 passes (4.6 coarse), team 4 scan 5.8 (2.8 coarse); a coarse pass
  ends on the true sign change (1.0 per period), on the curvature test just before it (0.8) or on a change of the
  layer dropping (0.06); the figures above are SURFDISP_EXACTSCAN."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "pysurfinv_amd", "lib", "libsurfdisp_stats.so")
from pysurfinv_amd import forward, synth
L = _lib.lib()
L.surfdisp_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
per = torch.from_numpy(synth.default_periods(20)).cuda()
model = torch.from_numpy(synth.synth_models(65536, 10, seed=0)).cuda()
plan = forward.BatchPlan(65536, 10, 20)
for team in (2, 4, 8):
    for fs in (True, False):
        L.surfdisp_set_team(team)
        L.surfdisp_stats(None, 1)
        plan.run(model, per, kind=2, exact_scan=fs); torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 24)()
        L.surfdisp_stats(out, 0)
        s = np.array(list(out), float)
        n = s[4]
        print(f"team {team} exact_scan {fs}: per solved period: scan passes {s[0]/n:.2f} (coarse {s[5]/n:.2f}, of which ended by an "
              f"uncertified interval {s[6]/n:.2f}), refine {s[1]/n:.2f}, ellip {s[2]/n:.2f}; evaluations/period ~ {(s[0]+s[1])*team/n + 2:.1f}"
              + (f"; coarse-pass endings per period: sign change {s[9]/n:.2f}, guard {s[10]/n:.2f}, near half space {s[11]/n:.2f}, "
                 f"layer dropping changed {s[12]/n:.2f}, log curvature {s[14]/n:.3f}, phase {s[15]/n:.3f}, "
                 f"entry slope {s[16]/n:.3f}, other {s[17]/n:.3f}; restarts one interval back {s[18]/n:.3f}" if s[5] else ""))
L.surfdisp_set_team(0)
