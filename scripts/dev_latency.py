#!/usr/bin/env python3
"""Latency of the one-stack drop-in call fast_surf() (compatibility path)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import fast_surf, synth
from oracle import refso
for L in (5, 10, 64, 96):
    m = synth.synth_models(1, L, seed=0)[0].astype(np.float64)
    per = np.zeros(200); per[:20] = synth.default_periods(20)
    for kind in (2, 1):
        fast_surf.fast_surf(L, kind, m[0], m[1], m[2], m[3], m[4], per, 20)
        t0 = time.perf_counter()
        for _ in range(50): fast_surf.fast_surf(L, kind, m[0], m[1], m[2], m[3], m[4], per, 20)
        dt = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(20): refso.fast_surf(L, kind, m[0], m[1], m[2], m[3], m[4], per, 20)
        dr = (time.perf_counter() - t0) / 20
        print(f"L={L:3d} kind={kind}: fast_surf() on MI355X {dt*1e3:7.3f} ms per call; reference Fortran on the host {dr*1e3:7.3f} ms", flush=True)
