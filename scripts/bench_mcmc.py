#!/usr/bin/env python3
"""Metropolis-driver throughput on one MI355X (BASELINE configs[2] and the per-GPU share of
configs[3]); prints one JSON line per configuration.  Continental 96-layer model of
tests/golden/settings.py, 19 periods, Rayleigh-only misfit (point.py:15-31)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
from settings import CONT
from pysurfinv_amd.layers_batch import Model1DBatch
from pysurfinv_amd.mcmc import MetropolisBatch

G = np.load(os.path.join(ROOT, "tests", "golden", "ref_driver.npz"))
dev = "cuda:0"
if os.environ.get("MCMC_MODEL", "cont") == "hybrid":
    # thermal oceanic model: 86 layers, parameters -> stack through thermseis (torch)
    from settings_therm import HYBRID_STATIC, PERIODS
    GT = np.load(os.path.join(ROOT, "tests", "golden", "ref_therm.npz"))
    mb = Model1DBatch(HYBRID_STATIC, device=dev)
    G = {"trace/periods": np.asarray(PERIODS, float), "trace/c_obs": GT["hyb_ritz/c"][0] * 1.002,
         "trace/uncer": np.full(len(PERIODS), 0.01)}
else:
    mb = Model1DBatch(CONT, device=dev)
IND = os.environ.get("MCMC_INDEPENDENT", "0") == "1"
XS = os.environ.get("MCMC_FASTSCAN", "0") == "1"
for name, chains, steps, depth in (("configs[2] single point: 100 chains (100 000 steps = 100 x 1000)", 100, 61, 1),
                                   ("configs[2], speculative depth 3", 100, 61, 3),
                                   ("configs[2], speculative depth 4", 100, 61, 4),
                                   ("configs[2], speculative depth 5", 100, 61, 5),
                                   ("1 024 chains", 1024, 41, 1),
                                   ("1 024 chains, speculative depth 2", 1024, 41, 2),
                                   ("configs[3] per-GPU share: 512 points x 50 chains = 25 600 chains", 25600, 12, 1)):
    mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"], device=dev, seed=0, independent=IND, fast_scan=XS)
    mc.run(chains, 1 + 2 * depth, spec_depth=depth); torch.cuda.synchronize()
    t0 = time.perf_counter(); tr = mc.run(chains, steps, spec_depth=depth); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if depth == 1 and os.environ.get("MCMC_GRAPH", "1") == "1":
        mc.run_graphed(chains, 8); torch.cuda.synchronize()
        t0 = time.perf_counter(); trg = mc.run_graphed(chains, 203); torch.cuda.synchronize(); dtg = time.perf_counter() - t0
        print(json.dumps({"independent": IND, "fast_scan": XS, "config": name + " [HIP graph]", "chains": chains,
                          "metropolis_steps_per_s": chains * 203 / dtg, "ms_per_lockstep": dtg / 203 * 1e3,
                          "ms_params_to_stack": 0, "ms_forward_phase_only": 0, "accept_rate": float(trg[:, 1:, 2].mean())}), flush=True)
    # split: parameters -> stacks, forward, rest
    p = mc.reset(chains); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5): m, nl = mb.to_model(p)
    torch.cuda.synchronize(); t_model = (time.perf_counter() - t1) / 5
    t1 = time.perf_counter()
    for _ in range(5): mc.forward_c(p)
    torch.cuda.synchronize(); t_fwd = (time.perf_counter() - t1) / 5 - t_model
    print(json.dumps({"independent": IND, "fast_scan": XS, "config": name, "chains": chains, "steps_timed": steps, "spec_depth": depth, "layers": int(m.shape[2]),
                      "metropolis_steps_per_s": chains * steps / dt, "ms_per_lockstep": dt / steps * 1e3,
                      "ms_params_to_stack": t_model * 1e3, "ms_forward_phase_only": t_fwd * 1e3,
                      "accept_rate": float(tr[:, 1:, 2].mean())}), flush=True)
