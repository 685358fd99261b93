#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

Runs only in the development container (needs /root/reference and the flang
build oracle/_ref/libfast_surf_ref.so made by oracle/build_ref.sh).  What is
committed is DATA: input layer stacks + period lists and the outputs the
reference Fortran produced for them, plus the reference's own known-answer
files senskernel-1.0/TEST1/{eus_model,test.[RL].{phv,grv}} parsed to arrays.

Every reference solve is made with fresh-process state (oracle/refso.py resets
ndiv and /dispe/; SURVEY.md section 4 defect 1), ascending periods.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refso                      # noqa: E402
from pysurfinv_amd.synth import synth_models, water_models, default_periods  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
T1 = os.path.join(REF, "senskernel-1.0", "TEST1")


def first_block(fn):
    rows = []
    for line in open(fn):
        s = line.split()
        if not s:
            if rows:
                break
            continue
        rows.append([float(x) for x in s])
    return np.array(rows)


def run_ref(model, periods, kind):
    c, u = refso.forward_batch(model[:, 0], model[:, 1], model[:, 2], model[:, 3], model[:, 4],
                               periods, kind)
    return c, u


def main():
    # ---- (i) the reference's own known-answer test ---------------------------
    eus = np.loadtxt(os.path.join(T1, "eus_model"))          # columns h, vp, vs, rho, Qs
    h, vp, vs, rho, qs = eus.T
    eus_model = np.stack([vp, vs, rho, h, 1.0 / qs])[None].astype(np.float32)
    per10 = np.arange(10.0, 101.0, 10.0).astype(np.float32)
    out = dict(model=eus_model, periods=per10)
    for w, kind in (("R", 2), ("L", 1)):
        phv = first_block(os.path.join(T1, f"test.{w}.phv"))
        grv = first_block(os.path.join(T1, f"test.{w}.grv"))
        assert np.allclose(phv[:, 0], per10) and np.allclose(grv[:, 0], per10)
        out[f"c_{w}_fp64twin"] = phv[:, 1]
        out[f"u_{w}_fp64twin"] = grv[:, 1]
        c, u = run_ref(eus_model, per10, kind)
        out[f"c_{w}_ref"] = c[0]
        out[f"u_{w}_ref"] = u[0]
    np.savez_compressed(os.path.join(HERE, "test1_eus.npz"), **out)

    # ---- (ii) captured outputs of the reference Fortran -----------------------
    P20 = default_periods(20)
    cases = {}

    def add(name, model, periods, kinds=(2, 1)):
        periods = np.asarray(periods, np.float32)
        for kind in kinds:
            c, u = run_ref(model, periods, kind)
            key = f"{name}_{'R' if kind == 2 else 'L'}"
            cases[key] = dict(model=model, periods=periods, kind=kind, c=c, u=u)
            print(f"{key:28s} B={model.shape[0]:3d} L={model.shape[2]:3d} P={len(periods):2d} "
                  f"solved={np.mean(c > 0):.3f}")

    add("c1_single_L5", synth_models(1, 5, 0), P20)                  # BASELINE config 1
    add("synth_L5", synth_models(32, 5, 0), P20)
    add("synth_L10", synth_models(64, 10, 0), P20)                   # BASELINE config 2 shape
    add("synth_L21", synth_models(16, 21, 1), P20)                   # ndiv clamps to 4 (R)
    add("synth_L64", synth_models(16, 64, 2), P20)                   # BASELINE config 5 shape
    add("rough_L10", synth_models(64, 10, 3, noise=0.15, monotone=False), P20)
    add("rough_L64", synth_models(32, 64, 3, noise=0.15, monotone=False), P20)
    add("water_L9", water_models(16), np.linspace(6, 80, 18))
    add("sparse_L10", synth_models(16, 10, 1), [5., 20., 40., 60., 80.])
    add("dense18_L10", synth_models(16, 10, 4), np.linspace(5, 90, 18))
    vs = np.array([3.2, 3.6, 4.5, 3.0]); hh = np.array([10, 20, 40, 0.])
    vp = 1.76 * vs; rho = 0.541 + 0.3601 * vp
    lvz = np.stack([vp, vs, rho, hh, np.full(4, 1 / 300.)])[None].astype(np.float32)
    add("lvz_halfspace_L4", lvz, np.linspace(6, 80, 18))
    two = synth_models(4, 2, 7)
    add("two_layer", two, P20)
    # thick rough stacks at short periods: the Rayleigh eigenfunction solutions grow by up to ~1e27
    # between the effective half space and the surface (found by scripts/soak.py)
    add("rough_thick_L6", synth_models(48, 6, 21, noise=0.2, monotone=False, total_thickness=400.), np.linspace(5, 40, 8))
    add("rough_thick_L12", synth_models(48, 12, 22, noise=0.2, monotone=False, total_thickness=340.), np.linspace(4, 60, 10))
    add("rough_thick_L22", synth_models(32, 22, 23, noise=0.2, monotone=False, total_thickness=400.), np.linspace(3.3, 50, 12))

    flat = {}
    for key, d in cases.items():
        for f, v in d.items():
            flat[f"{key}/{f}"] = np.asarray(v)
    flat["__meta__/build"] = np.array(open(os.path.join(ROOT, "oracle", "_ref", "BUILD_INFO.txt")).read())
    np.savez_compressed(os.path.join(HERE, "ref_cases.npz"), **flat)
    print("wrote", os.path.join(HERE, "ref_cases.npz"))


if __name__ == "__main__":
    main()
