#!/usr/bin/env python3
"""Where the time of the thermal parameters->stack path goes (development)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import thermseis as ts

dev = torch.device("cuda:0")
B, N = 16384, 61
torch.manual_seed(0)
age = torch.rand(B, dtype=torch.float64, device=dev) * 10 + 0.1
z = torch.linspace(0, 190, N, dtype=torch.float64, device=dev)[None, :].expand(B, N).contiguous()
ch = torch.full((B,), 4.4, dtype=torch.float64, device=dev)


def T(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms", flush=True)
    return r


th = T("hscm(age, [B,61])", lambda: ts.hscm(age, zdeps=z + 4.4))
T("hscm_mantle_temperature", lambda: ts.hscm_mantle_temperature(age[:, None]))
vs = T("ritz_vs", lambda: ts.ritz_vs(th)[0])
T("ruan", lambda: ts.ruan(th, 1))
zm = T("melt_start (hscm [B,200])", lambda: ts.melt_start(age, ch))
keep = (z < zm[:, None]) | (z > ((zm + ch) * 1.7 - ch)[:, None])
T("cubic_spline_through", lambda: ts.cubic_spline_through(z, vs, keep))
