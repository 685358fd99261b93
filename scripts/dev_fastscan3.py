#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward, synth
np.set_printoptions(linewidth=250, precision=4, suppress=True)
L = _lib.lib()
m = synth.synth_models(256, 7, seed=7, noise=0.05, monotone=True, total_thickness=400.0)
for T in (3.219, 4.0, 5.0, 8.0):
    per = np.array([T], np.float32)
    row = []
    for team in (2, 4, 8):
        L.surfdisp_set_team(team)
        c0, _, _ = forward.forward_batch(m, per, 1)
        c1, _, _ = forward.forward_batch(m, per, 1, fastscan=True)
        row.append((team, int((c0 != c1).sum()), float(c0[0, 0]), float(c1[0, 0])))
    print("T", T, row)
