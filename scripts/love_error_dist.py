#!/usr/bin/env python3
"""GPU box: distribution of the default root search's phase-velocity error against SURFDISP_STRICT on random stacks
(the generator of scripts/soak.py, one wave type): share of values beyond 1e-5 / 2e-5 / 5e-5 / 1e-4 and the worst."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward, synth
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(11)
bins = np.array([1e-5, 2e-5, 5e-5, 1e-4, 1e-3])
cnt = np.zeros(len(bins)); tot = 0; worst = 0.0; npat = 0; nst = 0
for it in range(ncase):
    L = int(rng.integers(2, 48)); B = int(rng.integers(2048, 16384))
    noise = float(rng.choice([0.02, 0.05, 0.1, 0.2])); mono = bool(rng.random() < 0.6)
    model = synth.synth_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise, monotone=mono,
                               total_thickness=float(rng.choice([60., 120., 200., 400.])))
    if os.environ.get("FAMILY") == "sediment" and L >= 4:
        model = synth.sediment_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise, total_thickness=float(rng.choice([30., 60., 120., 200.])))
    P = int(rng.integers(4, 40))
    per = np.sort(rng.uniform(3.0, 150.0, P)).astype(np.float32)
    if os.environ.get("FAMILY") == "sediment":
        per = np.sort(rng.uniform(0.3, 30.0, P)).astype(np.float32)
    _lib.lib().surfdisp_set_team(int(rng.choice([0, 2, 4, 8, 16])))
    c, u, st = forward.forward_batch(model, per, kind)
    co, uo, so = forward.forward_batch(model, per, kind, strict=True)
    rows = ((c > 0) == (co > 0)).all(axis=1)
    ok = (co > 0) & rows[:, None]
    e = np.abs(c[ok].astype(np.float64) / co[ok] - 1)
    cnt += [(e > b).sum() for b in bins]; tot += e.size; worst = max(worst, float(e.max()) if e.size else 0.0)
    npat += int((~rows).sum()); nst += B
_lib.lib().surfdisp_set_team(0)
print(f"kind {kind} family {os.environ.get('FAMILY', 'general')}: {ncase} cases, {nst} stacks, {tot} values; zero-pattern mismatches {npat} ({npat / nst:.2e}); "
      + "  ".join(f">{b:.0e}: {int(n)} ({n / tot:.2e})" for b, n in zip(bins, cnt)) + f"; worst {worst:.2e}")
