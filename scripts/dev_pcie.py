#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (DESIGN.md section 6 note)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import forward, synth
model = synth.synth_models(65536, 10, seed=0); per = synth.default_periods(20)
forward.forward_batch(model, per, 2)
t0 = time.perf_counter()
for _ in range(5): forward.forward_batch(model, per, 2)
dt = (time.perf_counter() - t0) / 5
print(f"host-buffer entry, B=65536 L=10 P=20 Rayleigh c+U: {dt*1e3:.2f} ms per call = {65536/dt/1e6:.2f} M solves/s (PCIe + hipMalloc/hipFree inclusive)")
