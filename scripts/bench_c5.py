#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: 64-layer ThermSeis-derived stacks (water + Cascadia sediment +
2-layer crust + 60-layer OceanMantleHybrid), joint Rayleigh + Love phase + group velocity at 20
periods, and finite-difference Vs sensitivity kernels of every stack (2 x 64 + 1 solves per stack and
wave type).  Prints one JSON line per stage."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
from pysurfinv_amd import forward, senskernel, synth            # noqa: E402
from pysurfinv_amd.brownian import TorchProposer                # noqa: E402
from pysurfinv_amd.layers_batch import Model1DBatch             # noqa: E402

SETTING = {
    'OceanWater': {'H': 2.6},
    'OceanSedimentCascadia': {'H': [0.3, 'abs', 0.2, 0.03]},
    'OceanCrust': {'H': 4.4, 'Vs': [3.25, 3.94]},
    'OceanMantleHybrid': {'BottomDepth': 200, 'Conversion': 'Ritzwoller', 'ThermAge': [4, 'rel_pos', 200, 0.4],
                          'Vs': [[0, 'abs', 0.2, 0.01], [0, 'abs', 0.2, 0.01], [0, 'abs', 0.2, 0.01], [0, 'abs', 0.1, 0.01]]},
    'Info': {'modelType': 'MCInv', 'period': 10, 'refLayer': False},
}


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


def main():
    dev = torch.device("cuda:0")
    B = int(os.environ.get("C5_STACKS", "16384"))
    MS = int(os.environ.get("C5_SENS_STACKS", "1024"))
    per = torch.from_numpy(synth.default_periods(20)).to(dev)
    mb = Model1DBatch(SETTING, device=dev)
    params = TorchProposer(mb.spec, dev, seed=1).reset(B)
    dt, (model_t, nlay_t) = timed(lambda: mb.to_model_torch(params), 3)
    L = model_t.shape[2]
    print(json.dumps({"stage": "thermal parameters -> layer stack (torch mirror, device)", "stacks": B, "layers": L,
                      "ms": dt * 1e3, "stacks_per_s": B / dt}), flush=True)
    dt, (model, nlay) = timed(lambda: mb.to_model(params), 10)
    err = float(((model - model_t).abs() / model_t.abs().clamp(min=1e-3)).max())
    print(json.dumps({"stage": "thermal parameters -> layer stack (HIP: surfdisp_thermal_kernel + surfdisp_layers_kernel)",
                      "stacks": B, "layers": L, "ms": dt * 1e3, "stacks_per_s": B / dt,
                      "max_rel_diff_vs_torch_mirror": err}), flush=True)
    plan = forward.JointPlan(B, L, 20, device=dev)
    dt, out = timed(lambda: plan.run(model, per, nlay=nlay), 5)
    okR = float((out["statusR"] == 0).float().mean()); okL = float((out["statusL"] == 0).float().mean())
    print(json.dumps({"stage": "joint Rayleigh+Love c+U", "stacks": B, "layers": L, "ms": dt * 1e3,
                      "stacks_per_s": B / dt, "solves_per_s": 2 * B / dt, "solved_R": okR, "solved_L": okL}), flush=True)
    for w in ("R", "L"):
        dt, k = timed(lambda: senskernel.analytic_kernels(model, per, wtype=w, nlay=nlay), 3)
        print(json.dumps({"stage": f"analytic dc/dVs, dc/dVp, dc/drho kernels of every layer (one solve), {w}", "stacks": B,
                          "layers": L, "ms": dt * 1e3, "kernel_sets_per_s": B / dt}), flush=True)
    for w in ("R", "L"):
        dt, k = timed(lambda: senskernel.sens_kernel_pert_batch(model[:MS], per, wtype=w, nlay=None if nlay is None else nlay[:MS]), 2)
        nan = float(torch.isnan(k["phv"]).float().mean())
        print(json.dumps({"stage": f"finite-difference Vs sensitivity kernels, {w}", "stacks": MS, "layers": L,
                          "solves": MS * (2 * L + 1), "ms": dt * 1e3, "kernel_sets_per_s": MS / dt,
                          "solves_per_s": MS * (2 * L + 1) / dt, "nan_fraction": nan}), flush=True)


if __name__ == "__main__":
    main()
