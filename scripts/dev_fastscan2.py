#!/usr/bin/env python3
"""Development: where does SURFDISP_FASTSCAN differ from the faithful scan?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward, synth
from oracle import cport
np.set_printoptions(linewidth=250, precision=4, suppress=True)
L = _lib.lib()
rng = np.random.default_rng(1)
shown = 0
stat = {}
for it in range(int(os.environ.get("FS_CASES", "60"))):
    Ln = int(rng.integers(3, 40)); B = 2048; kind = int(rng.integers(1, 3))
    noise = float(rng.choice([0.02, 0.05, 0.1, 0.2])); mono = bool(rng.random() < 0.5); tt = float(rng.choice([60., 120., 200.]))
    m = synth.synth_models(B, Ln, seed=int(rng.integers(1 << 30)), noise=noise, monotone=mono, total_thickness=tt)
    P = int(rng.integers(5, 30)); per = np.sort(rng.uniform(4.0, 120.0, P)).astype(np.float32)
    team = int(rng.choice([2, 4, 8])); L.surfdisp_set_team(team)
    c0, u0, s0 = forward.forward_batch(m, per, kind)
    c1, u1, s1 = forward.forward_batch(m, per, kind, fastscan=True)
    bad = np.nonzero((c0 != c1).any(axis=1))[0]
    key = (noise, mono)
    stat[key] = stat.get(key, 0) + bad.size
    if bad.size and shown < 6:
        shown += 1
        i = bad[0]
        co, uo, so = cport.forward_batch(m[i:i + 1], per, kind)
        k = int(np.nonzero(c0[i] != c1[i])[0][0])
        print(f"L={Ln} kind={kind} noise={noise} mono={mono} thick={tt} team={team}: {bad.size} stacks differ; stack {i}, first differing period {k} (T={per[k]:.2f})")
        print("  faithful", c0[i][max(0, k - 2):k + 3], s0[i]); print("  fast    ", c1[i][max(0, k - 2):k + 3], s1[i]); print("  oracle  ", co[0][max(0, k - 2):k + 3], so[0])
        print("  vs", m[i, 1].round(2).tolist())
print({k: v for k, v in sorted(stat.items())})
