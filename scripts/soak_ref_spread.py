#!/usr/bin/env python3
"""How often does the REFERENCE's own arithmetic disagree with itself on the zero pattern?  CPU only.
The same random families as scripts/soak.py, solved by two builds of the C restatement of the reference
(oracle/surfdisp_oracle.c): -ffp-contract=off (the committed oracle, bit-exact with the reference Fortran built the same
way) and -ffp-contract=fast -march=native (what an optimising build of the Fortran does to the same statements).
The fraction of stacks whose zero pattern differs between the two is the yardstick for the GPU soaks' mismatch rates
(scripts/soak.py): a scan step within rounding of a root is decided by the last bit of the secular function.
    SOAK_SECONDS=600 SOAK_FAMILY=general|sediment|overflow python scripts/soak_ref_spread.py"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import synth                           # noqa: E402  (pure numpy generators)
from oracle import cport                                  # noqa: E402

FMA_SO = "/tmp/libsurfdisp_oracle_fma.so"
subprocess.check_call(["gcc", "-O2", "-ffp-contract=fast", "-march=native", "-fPIC", "-fopenmp", "-shared", "-o", FMA_SO,
                       os.path.join(ROOT, "oracle", "surfdisp_oracle.c"), "-lm"])


LIB_OFF = cport.lib()                                     # the committed oracle (contraction off)
cport._SO, cport._lib = FMA_SO, None
LIB_FMA = cport.lib()                                     # same source, contraction on (argtypes set by cport.lib)
cport._lib = LIB_OFF


def forward_with(lib, model, per, kind, nthreads):
    cport._lib = lib
    try:
        return cport.forward_batch(model, per, kind, nthreads=nthreads)
    finally:
        cport._lib = LIB_OFF


def main():
    rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "0")))
    t_end = time.time() + float(os.environ.get("SOAK_SECONDS", "120"))
    fam = os.environ.get("SOAK_FAMILY", "general")
    nthreads = int(os.environ.get("SOAK_THREADS", str(os.cpu_count() or 8)))
    nstack = npat = ncase = 0
    worst_c = 0.0
    t_last = time.time()
    while time.time() < t_end:
        L = int(rng.integers(2, 48)); B = int(rng.integers(256, 2048)); kind = int(rng.integers(1, 3))
        noise = float(rng.choice([0.02, 0.05, 0.1, 0.2])); mono = bool(rng.random() < 0.6)
        model = synth.synth_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise, monotone=mono,
                                   total_thickness=float(rng.choice([60., 120., 200., 400.])))
        P = int(rng.integers(1, 40))
        per = np.sort(rng.uniform(3.0, 150.0, P)).astype(np.float32)
        if fam == "sediment" and L >= 4:
            model = synth.sediment_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise,
                                          total_thickness=float(rng.choice([30., 60., 120., 200., 400.])))
            per = np.sort(rng.uniform(0.3, 30.0, P)).astype(np.float32)
        if fam == "overflow":
            L = int(rng.integers(2, 6))
            model = synth.synth_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise, monotone=mono,
                                       total_thickness=float(rng.uniform(60., 250.)) * L)
            per = np.sort(rng.uniform(2.5, 40.0, P)).astype(np.float32)
        c0, u0, s0 = forward_with(LIB_OFF, model, per, kind, nthreads)
        c1, u1, s1 = forward_with(LIB_FMA, model, per, kind, nthreads)
        rows = ((c0 > 0) == (c1 > 0)).all(axis=1)
        ok = (c0 > 0) & rows[:, None]
        if ok.any():
            worst_c = max(worst_c, float(np.abs(c1[ok] / c0[ok] - 1).max()))
        nstack += B; ncase += 1; npat += int((~rows).sum())
        if time.time() - t_last > 60:
            t_last = time.time()
            print(f"  ... {ncase} cases, {nstack} stacks, pattern differences {npat}", flush=True)
    print(f"reference arithmetic against itself ({fam}; FMA contraction off vs on): {ncase} cases, {nstack} stacks, "
          f"zero-pattern differences {npat} stacks ({npat / max(nstack, 1):.2e}), worst c difference where both solved {worst_c:.2e}")


if __name__ == "__main__":
    main()
