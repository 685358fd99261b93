#!/usr/bin/env python3
"""Development: two- and three-layer stacks with very thick layers (soak findings)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward, synth
from oracle import cport
np.set_printoptions(linewidth=250, precision=4, suppress=True)
rng = np.random.default_rng(5)
shown = 0
for L in (2, 3):
    for tt in (200., 400.):
        for noise in (0.05, 0.2):
            for rep in range(3):
                m = synth.synth_models(512, L, seed=int(rng.integers(1 << 30)), noise=noise, monotone=bool(rep % 2), total_thickness=tt)
                per = np.sort(rng.uniform(3.0, 150.0, 31)).astype(np.float32)
                c, u, st = forward.forward_batch(m, per, 2)
                co, uo, so = cport.forward_batch(m, per, 2, nthreads=16)
                rows = ((c > 0) == (co > 0)).all(axis=1)
                bad = np.nonzero(~rows)[0]
                pairs = sorted(set(zip(st[bad].tolist(), so[bad].tolist())))
                print(f"L={L} thick={tt} noise={noise} T0={per[0]:.2f}: mismatch {bad.size}/512 pairs {pairs} gpu st {np.bincount(st, minlength=9).tolist()} oracle {np.bincount(so, minlength=5).tolist()}")
                if bad.size > 50 and shown < 3:
                    shown += 1
                    i = bad[0]
                    print(" per     ", per); print(" model", m[i].round(3).tolist(), "st", st[i], so[i])
                    print(" c gpu   ", c[i]); print(" c oracle", co[i])
