#!/usr/bin/env python3
"""How far the reference moves from ITSELF: every golden case of ref_cases.npz run again through a second build
of the same unmodified Fortran with FMA contraction (oracle/_ref/libfast_surf_ref_fma.so, made by
oracle/build_ref.sh: flang -O2 -ffp-contract=fast -march=native; the golden vectors come from the
-ffp-contract=off build).  Development container only (needs /root/reference).  Writes

  tests/golden/ref_spread.npz    <case>/c_fma, <case>/u_fma            (data)
  tests/golden/u_exceptions.json the (case, stack, period) entries at which the reference's own group velocity
                                 differs between its two builds by more than 2e-5 relative or is NaN in one of
                                 them - entries next to osculating modes, where U changes by >1000x the relative
                                 change of c; tests/test_gpu_parity.py holds every OTHER entry to the 1e-4 bar.

    python tests/golden/make_golden_spread.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cases                 # noqa: E402
from oracle import refso                        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SPREAD_LIST = 2e-5


def main():
    refso._SO = os.path.join(ROOT, "oracle", "_ref", "libfast_surf_ref_fma.so")
    refso._lib = None
    out, exc = {}, []
    for name, d in sorted(load_cases().items()):
        m = d["model"]
        c, u = refso.forward_batch(m[:, 0], m[:, 1], m[:, 2], m[:, 3], m[:, 4], d["periods"], d["kind"])
        out[f"{name}/c_fma"], out[f"{name}/u_fma"] = c, u
        assert np.array_equal(c > 0, d["c"] > 0), f"{name}: the two builds of the reference disagree on the zero pattern"
        ok = d["c"] > 0
        with np.errstate(invalid="ignore", divide="ignore"):
            su = np.abs(u.astype(np.float64) / d["u"] - 1.0)
            sc = np.abs(c.astype(np.float64) / d["c"] - 1.0)
        bad = ok & ~(su <= SPREAD_LIST)                        # NaN counts
        print(f"{name:22s} c spread max {np.nanmax(np.where(ok, sc, 0)):.1e}   U spread max "
              f"{np.nanmax(np.where(ok & np.isfinite(su), su, 0)):.1e}   listed {int(bad.sum())}")
        for b, k in zip(*np.where(bad)):
            exc.append(dict(case=name, stack=int(b), period_index=int(k), T=float(d["periods"][k]),
                            c_ref=float(d["c"][b, k]), u_ref=float(d["u"][b, k]),
                            c_ref_fma=float(c[b, k]), u_ref_fma=(float(u[b, k]) if np.isfinite(u[b, k]) else "NaN"),
                            ref_spread_c=float(sc[b, k]), ref_spread_u=(float(su[b, k]) if np.isfinite(su[b, k]) else "NaN")))
    np.savez_compressed(os.path.join(HERE, "ref_spread.npz"), **out)
    with open(os.path.join(HERE, "u_exceptions.json"), "w") as f:
        json.dump({"made_by": "tests/golden/make_golden_spread.py",
                   "builds": "flang -O2 -ffp-contract=off (golden) vs flang -O2 -ffp-contract=fast -march=native",
                   "listed_when": f"reference's own U spread > {SPREAD_LIST} relative, or NaN in one build",
                   "entries": exc}, f, indent=1)
    print("listed", len(exc), "entries")


if __name__ == "__main__":
    main()
