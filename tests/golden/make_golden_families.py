#!/usr/bin/env python3
"""One small committed fixture per family of the randomised differential soaks (scripts/soak.py,
scripts/soak_scan.py), so that the driver's `pytest -m gpu` run sees them: input stacks + period lists and what the
reference Fortran (oracle/_ref/libfast_surf_ref.so, fresh-process state, oracle/refso.py) returns for them.
Development container only.  Families:

  sediment   soft sediments (Vs 0.2-1.4 km/s, Vp/Vs 1.8-3.5) over rock, periods 0.3-30 s
  wild       anything monotone: Vs 0.1-5 km/s, thicknesses 10 m-50 km, periods 0.1-300 s log-uniform, Vp/Vs 1.5-8
             (with layers of 0.1-0.3 km/s: where the opt-in fast scan was seen to differ from the point-by-point scan)
  overflow   two to four layers of 100-200 km at periods of 3-8 s: the un-normalised fp32 secular function overflows
  ragged     rough stacks (sigma 0.2, unsorted) with 2-47 layers, some under water + sediment, random period lists

    python tests/golden/make_golden_families.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refso                                    # noqa: E402
from pysurfinv_amd import synth                             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run_ref(model, nlay, periods, kind):
    B, _, L = model.shape
    P = len(periods)
    c = np.zeros((B, P), np.float32); u = np.zeros((B, P), np.float32)
    for i in range(B):
        n = int(nlay[i])
        m = model[i, :, :n]
        ur, ul, cr, cl = refso.fast_surf(n, kind, m[0], m[1], m[2], m[3], m[4], periods, P)
        c[i], u[i] = (cr[:P], ur[:P]) if kind == 2 else (cl[:P], ul[:P])
    return c, u


def wild(B, L, rng, vsmin=0.1):
    m = synth.synth_models(B, L, seed=int(rng.integers(1 << 30)))
    vs = np.sort(np.exp(rng.uniform(np.log(vsmin), np.log(5.0), (B, L))), axis=1)
    vp = np.sort(vs * np.exp(rng.uniform(np.log(1.5), np.log(8.0), (B, L))), axis=1)
    m[:, 1, :] = vs; m[:, 0, :] = vp; m[:, 2, :] = np.sort(rng.uniform(1.6, 3.4, (B, L)), axis=1)
    m[:, 3, :] = np.exp(rng.uniform(np.log(0.01), np.log(50.0), (B, L))); m[:, 3, -1] = 0.0
    m[:, 4, :] = rng.choice([1e-4, 1 / 600., 1 / 80., 1 / 20.], (B, L))
    return m.astype(np.float32)


def main():
    rng = np.random.default_rng(2026)
    fam = {}
    full = lambda m: np.full(m.shape[0], m.shape[2], np.int32)
    # sediment
    m = np.concatenate([synth.sediment_models(24, 12, seed=5, noise=0.1, total_thickness=120.0),
                        synth.sediment_models(24, 12, seed=6, noise=0.02, total_thickness=30.0, max_layers=12)])
    fam["sediment"] = (m, full(m), np.sort(rng.uniform(0.3, 30.0, 16)).astype(np.float32))
    # wild, slow top layers
    m = wild(64, 8, rng)
    fam["wild"] = (m, full(m), np.sort(np.exp(rng.uniform(np.log(0.1), np.log(300.0), 16))).astype(np.float32))
    # overflow regime
    m = np.concatenate([synth.synth_models(16, 2, seed=2, noise=0.05, monotone=False, total_thickness=400.0)[:, :, [0, 1, 1, 1]],
                        synth.synth_models(16, 4, seed=4, noise=0.1, monotone=False, total_thickness=500.0)])
    nl = np.concatenate([np.full(16, 2), np.full(16, 4)]).astype(np.int32)
    fam["overflow"] = (m, nl, np.linspace(3.0, 40.0, 12).astype(np.float32))
    # ragged rough
    B, Lmax = 64, 47
    nl = rng.integers(2, Lmax + 1, B).astype(np.int32)
    m = np.zeros((B, 5, Lmax), np.float32)
    for i, n in enumerate(nl):
        s = synth.synth_models(1, int(n), seed=3000 + i, noise=0.2, monotone=False,
                               total_thickness=float(rng.choice([60., 120., 200., 400.])))[0]
        if n >= 4 and rng.random() < 0.3:                   # water + thin sediment on top
            s[1, 0] = 0.0; s[0, 0] = 1.475; s[2, 0] = 1.027; s[4, 0] = 1e-4; s[3, 0] = rng.uniform(0.3, 4.0)
            s[1, 1] = 1.0; s[0, 1] = 2.5; s[2, 1] = 2.0; s[3, 1] = rng.uniform(0.2, 1.0)
        m[i, :, :n] = s
    fam["ragged"] = (m, nl, np.sort(rng.uniform(4.0, 120.0, 14)).astype(np.float32))

    flat = {}
    for name, (m, nl, per) in fam.items():
        for kind in (2, 1):
            c, u = run_ref(m, nl, per, kind)
            key = f"{name}_{'R' if kind == 2 else 'L'}"
            flat[f"{key}/model"] = m; flat[f"{key}/nlay"] = nl; flat[f"{key}/periods"] = per
            flat[f"{key}/kind"] = np.int32(kind); flat[f"{key}/c"] = c; flat[f"{key}/u"] = u
            print(f"{key:14s} B={m.shape[0]} Lmax={m.shape[2]} P={len(per)}  periods solved {np.mean(c > 0):.3f}  "
                  f"stacks fully solved {np.mean((c > 0).all(1)):.3f}  non-finite U {int((~np.isfinite(u)).sum())}")
    flat["__meta__/build"] = np.array(open(os.path.join(ROOT, "oracle", "_ref", "BUILD_INFO.txt")).read())
    np.savez_compressed(os.path.join(HERE, "ref_families.npz"), **flat)


if __name__ == "__main__":
    main()
