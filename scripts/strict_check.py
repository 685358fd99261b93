#!/usr/bin/env python3
"""Development check of the verification mode (SURFDISP_STRICT: every stack through the statement-by-statement kernel):
  (1) against the golden vectors captured from the reference Fortran and the soak families (tests/golden): how many
      phase velocities are bit-identical, worst relative difference, zero patterns;
  (2) default mode against strict mode on the bench batch (65 536 x L10 x P20) and on the grid-leg shape - a
      GPU-only differential at full size, no CPU oracle in the loop.
Run on the GPU box:  python scripts/strict_check.py > gpurun_out/strict_check.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_cases, load_families            # noqa: E402
from pysurfinv_amd import forward, synth                   # noqa: E402


def stats(name, c, ref, u=None, uref=None):
    c = np.asarray(c, np.float32); ref = np.asarray(ref, np.float32)
    both = (c > 0) & (ref > 0)
    pat = int(np.sum((c > 0) != (ref > 0)))
    same = int(np.sum(c[both] == ref[both]))
    e = np.abs(c[both].astype(np.float64) / ref[both] - 1.0) if both.any() else np.zeros(1)
    line = f"{name:34s} n={both.sum():8d} identical {same / max(both.sum(), 1):7.4f}  max|dc/c| {e.max():.2e}  pattern mismatches {pat}"
    if u is not None:
        u = np.asarray(u, np.float64); uref = np.asarray(uref, np.float64)
        ok = both & (uref != 0) & np.isfinite(uref)
        eu = np.abs(u[ok] / uref[ok] - 1.0) if ok.any() else np.zeros(1)
        eu = np.where(np.isfinite(eu), eu, np.inf)
        line += f"  max|dU/U| {eu.max():.2e} (99.9% {np.quantile(eu, 0.999):.2e})"
    print(line, flush=True)


def main():
    print("(1) strict mode vs the reference Fortran's golden vectors")
    for name, d in sorted(load_cases().items()):
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"], strict=True)
        stats(name, c, d["c"], u, d["u"])
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"])
        stats("   default mode", c, d["c"], u, d["u"])
    print("(1b) soak families (reference Fortran outputs)")
    for name, d in sorted(load_families().items()):
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"], strict=True)
        stats(name, c, d["c"], u, d["u"])
        c, u, st = forward.forward_batch(d["model"], d["periods"], d["kind"])
        stats("   default mode", c, d["c"], u, d["u"])
    print("(2) default mode vs strict mode, full size")
    per = synth.default_periods(20)
    for label, B, L, kind in (("bench batch 65536 x L10 R", 65536, 10, 2), ("65536 x L10 L", 65536, 10, 1),
                              ("16384 x L64 R", 16384, 64, 2)):
        m = synth.synth_models(B, L, seed=1)
        t0 = time.time(); cs, us, _ = forward.forward_batch(m, per, kind, strict=True); ts = time.time() - t0
        t0 = time.time(); cd, ud, _ = forward.forward_batch(m, per, kind); td = time.time() - t0
        stats(label, cd, cs, ud, us)
        print(f"      host-to-host seconds: strict {ts:.3f}, default {td:.3f}")


if __name__ == "__main__":
    main()
