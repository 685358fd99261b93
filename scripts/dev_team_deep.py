#!/usr/bin/env python3
"""Development: root-search time on deep stacks (continental prior draws, 96 layers, 19 periods, phase only) by team size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from settings import CONT
from pysurfinv_amd import _lib, forward
from pysurfinv_amd.layers_batch import Model1DBatch
from pysurfinv_amd.mcmc import MetropolisBatch
G = np.load(os.path.join(ROOT, "tests", "golden", "ref_driver.npz"))
mb = Model1DBatch(CONT, device="cuda:0")
mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"], device="cuda:0", seed=0)
per = torch.from_numpy(np.asarray(G["trace/periods"], np.float32)).cuda()
L = _lib.lib()
for B in (25600, 4096):
    md, nl = mb.to_model(mc.reset(B)); md = md.contiguous()
    plan = forward.BatchPlan(B, md.shape[2], per.numel())
    ref = None
    for team in (0, 4, 8, 16, 32):
        L.surfdisp_set_team(team)
        for xs in (False, True):
            c, u, s = plan.run(md, per, kind=2 | 0x10, nlay=nl, exact_scan=xs); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): plan.run(md, per, kind=2 | 0x10, nlay=nl, exact_scan=xs)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            if ref is None: ref = c.clone()
            print(f"B={B} L={md.shape[2]} team={team} (used {L.surfdisp_get_team(B, md.shape[2])}) exact_scan={xs}: {dt*1e3:7.2f} ms  max|dc/c| vs first {float(((c-ref).abs()/ref.clamp_min(1e-9)).max()):.1e} solved {float((s==0).float().mean()):.3f}", flush=True)
L.surfdisp_set_team(0)
