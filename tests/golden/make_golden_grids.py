#!/usr/bin/env python3
"""Golden vectors for ``Model1D.seisPropGrids`` (models.py:72-91) from the imported reference Python (development container
only; see _refimport.py): depth / Vs / Vp / rho / Qs / Qp at the grid points of every layer - interface points doubled, as
the reference returns them - and the group of each point, for parameter vectors drawn by the reference's own
``MCinv.reset()``; a continental setting (with the reference mantle), an oceanic one and the per-point ('topo') case.

    python tests/golden/make_golden_grids.py        ->  tests/golden/ref_grids.npz
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.install()
from pySurfInv.models import buildModel1D            # noqa: E402
from settings import CONT, OCEAN                     # noqa: E402

GROUPS = ["water", "sediment", "crust", "mantle"]
ZDEPS = np.concatenate([np.linspace(-1.0, 60.0, 62), np.linspace(62.0, 260.0, 34)])


def capture(setting, nsamp, seed, local=None):
    random.seed(seed)
    mod0 = buildModel1D(setting, local) if local is not None else buildModel1D(setting)
    ref = setting['Info'].get('refLayer', False)
    mods = [mod0] + [mod0.reset() for _ in range(nsamp - 1)]
    P, Gd, NG, V, M = [], [], [], [], []
    for m in mods:
        P.append(m._brownians())
        z, vs, vp, rho, qs, qp, grp = m.seisPropGrids(refLayer=ref)
        Gd.append(np.array([z, vs, vp, rho, qs, qp, [GROUPS.index(g) for g in grp]], float)); NG.append(len(z))
        V.append(m.value(ZDEPS)); M.append(m.moho())       # models.py:104-112 (both on seisPropGrids() WITHOUT the reference mantle)
    arr = np.zeros((nsamp, 7, max(NG)))
    for i, a in enumerate(Gd):
        arr[i, :, :a.shape[1]] = a
    import json
    yml = json.dumps([mods[0].toYML(), mods[3].toYML()])       # Model1D.toYML (models.py:60-70): what Point.MCinv stores as `setting`
    return dict(params=np.array(P), grids=arr, ngrid=np.array(NG), value=np.array(V), moho=np.array(M), toyml=np.array(yml))


def main():
    out = {}
    for name, setting in (("cont", CONT), ("ocean", OCEAN)):
        d = capture(setting, 12, seed=21)
        for k, v in d.items():
            out[f"{name}/{k}"] = v
        print(name, d["grids"].shape, np.unique(d["ngrid"]))
    # PureGird (models.py:163-186): a model frozen as grid profiles - the continental start model with its reference mantle
    # (doubled points at every interface) and a copy WITHOUT the doubled points (the pieces then close up at the boundaries)
    from pySurfInv.models import PureGird
    prof = buildModel1D(CONT).seisPropGrids(refLayer=True)
    z = np.asarray(prof[0]); keep = np.concatenate([[True], np.diff(z) > 0])
    for tag, pr in (("pg", prof), ("pg_nodup", tuple(np.asarray(a)[keep] for a in prof[:6]) + ([g for g, kk in zip(prof[6], keep) if kk],))):
        pg = PureGird(pr, info={})
        g = pg.seisPropGrids()
        L = pg.seisPropLayers()
        out[f"{tag}/in"] = np.array([np.asarray(a, float) for a in pr[:6]])
        out[f"{tag}/in_grp"] = np.array([GROUPS.index(x) for x in pr[6]])
        out[f"{tag}/grids"] = np.array([np.asarray(a, float) for a in g[:6]])
        out[f"{tag}/grids_grp"] = np.array([GROUPS.index(x) for x in g[6]])
        out[f"{tag}/layers"] = np.array([np.asarray(a, float) for a in L[:6]])
        out[f"{tag}/value"] = pg.value(ZDEPS)
        out[f"{tag}/moho"] = pg.moho()
        out[f"{tag}/c"] = np.asarray(pg.forward([8, 12, 20, 30, 45, 60, 80]))
        print(tag, out[f"{tag}/grids"].shape, out[f"{tag}/layers"].shape, out[f"{tag}/c"])
    out["groups"] = np.array(GROUPS)
    out["zdeps"] = ZDEPS
    np.savez_compressed(os.path.join(HERE, "ref_grids.npz"), **out)


if __name__ == "__main__":
    main()
