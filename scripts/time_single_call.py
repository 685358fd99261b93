#!/usr/bin/env python3
"""GPU box: latency of ONE drop-in call fast_surf(...) (BASELINE configs[0]: one stack through the Fortran-ABI symbol, host
arrays in and out).  (The reference Fortran's own time per call is bench.py's cpu_baseline: 0.6 ms at L = 10, Rayleigh c+U.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pysurfinv_amd import fast_surf as fs, synth
per = np.zeros(200, np.float32); per[:20] = synth.default_periods(20)
for L in (5, 10, 64, 96):
    m = synth.synth_models(1, L, seed=0, **({} if L == 10 else {"total_thickness": 220.0}))[0]
    for kind, name in ((2, "Rayleigh"), (1, "Love")):
        args = (L, kind, m[0], m[1], m[2], m[3], m[4], per, 20)
        fs.fast_surf(*args)
        K = 200
        t0 = time.perf_counter()
        for _ in range(K):
            out = fs.fast_surf(*args)
        dt = (time.perf_counter() - t0) / K
        line = f"L={L:3d} {name:8s}: {dt*1e6:7.1f} us per call"
        print(line, flush=True)
