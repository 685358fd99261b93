#!/usr/bin/env python3
"""Captures golden vectors for the rows next to the hot path from the
imported reference Python (development container only; see _refimport.py):

  * ``BsplBasis`` matrices (layers.py:4-45) for the (N, nBasis) pairs the layer classes use;
  * ``Model1D.seisPropLayers(refLayer)`` outputs (models.py:93-102) for parameter vectors drawn by
    the reference's own ``MCinv.reset()`` - a continental and an oceanic setting;
  * a ``Point.MCinv`` trace (point.py:32-89) with a fixed seed: mcTrack + obs.

    python tests/golden/make_golden_driver.py
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
from pySurfInv.layers import BsplBasis               # noqa: E402
from pySurfInv.point import Point                    # noqa: E402

CONT = {
    'Sediment': {'H': [2., 'abs_pos', 1.5, 0.1], 'Vs': [[1.5, 'abs', 0.5, 0.05], [2.2, 'abs', 0.5, 0.05]]},
    'Crust': {'H': [35., 'abs', 10., 1.0],
              'Vs': [[3.4, 'abs', 0.3, 0.02], [3.6, 'abs', 0.3, 0.02], [3.8, 'abs', 0.3, 0.02], [3.9, 'abs', 0.3, 0.02]]},
    'Mantle': {'H': 160., 'Vs': [[4.4, 'abs', 0.4, 0.02], [4.35, 'abs', 0.4, 0.02], [4.4, 'abs', 0.4, 0.02],
                                 [4.5, 'abs', 0.4, 0.02], [4.6, 'abs', 0.4, 0.02]]},
    'Info': {'modelType': 'MCInv', 'refLayer': True},
}
OCEAN = {
    'OceanWater': {'H': 2.5},
    'OceanSediment': {'H': [0.4, 'rel_pos', 100, 0.05], 'Vs': [1.0, 0.5, 1.6, 0.05]},
    'OceanCrust': {'H': [7., 'abs', 2.5, 0.2], 'Vs': [[3.25, 'abs', 0.3, 0.02], [3.94, 'abs', 0.3, 0.02]]},
    'OceanMantle': {'BottomDepth': [200., 'abs', 30., 2.0],
                    'Vs': [[4.4, 'abs', 0.4, 0.02], [4.2, 'abs', 0.4, 0.02], [4.3, 'abs', 0.4, 0.02], [4.5, 'abs', 0.4, 0.02]]},
    'Info': {'modelType': 'MCInv', 'refLayer': False},
}
PERIODS = [8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 50, 60, 70, 80]


def capture_layers(setting, nsamp, seed):
    random.seed(seed)
    mod0 = buildModel1D(setting)
    ref = setting['Info'].get('refLayer', False)
    P, H, NL, C = [], [], [], []
    mods = [mod0] + [mod0.reset() for _ in range(nsamp - 1)]
    for m in mods:
        P.append(m._brownians())
        out = m.seisPropLayers(refLayer=ref)[:-1]
        H.append(np.array(out)); NL.append(len(out[0]))
    Lmax = max(NL)
    arr = np.zeros((nsamp, 6, Lmax))
    for i, a in enumerate(H):
        arr[i, :, :a.shape[1]] = a
    for m in mods[:6]:
        c = m.forward(PERIODS)
        C.append(np.zeros(len(PERIODS)) if c is None else np.array(c))
    return dict(params=np.array(P), layers=arr, nlay=np.array(NL), c=np.array(C))


def main():
    out = {}
    for N, nb, deg in ((15, 4, None), (30, 5, None), (60, 5, None), (10, 3, None), (5, 4, None), (60, 6, None), (30, 4, 3)):
        z = np.linspace(0, 37.5, N + 1)
        out[f"bspl/{N}_{nb}_{deg}"] = BsplBasis(z, nb, deg).basis
    for name, setting in (("cont", CONT), ("ocean", OCEAN)):
        d = capture_layers(setting, 40, seed=11)
        for k, v in d.items():
            out[f"{name}/{k}"] = v
        print(name, "layers", d["layers"].shape, "nlay", np.unique(d["nlay"]), "npar", d["params"].shape[1])
    # Metropolis trace: observations = forward of a perturbed model + 0.5 % so that both accepts and
    # rejects occur
    random.seed(5)
    truth = buildModel1D(CONT).reset()
    c_obs = np.array(truth.forward(PERIODS)) * 1.003
    unc = np.full(len(PERIODS), 0.02)
    p = Point(CONT, periods=PERIODS, vels=list(c_obs), uncers=list(unc))
    os.makedirs("/tmp/refenv_mc", exist_ok=True)
    p.MCinv(outdir="/tmp/refenv_mc", pid="trace", runN=240, chainL=80, seed=7)
    mc = np.load("/tmp/refenv_mc/trace.npz", allow_pickle=True)["mcTrack"]
    out["trace/mcTrack"] = mc
    out["trace/c_obs"] = c_obs
    out["trace/uncer"] = unc
    out["trace/periods"] = np.array(PERIODS, float)
    out["trace/meta"] = np.array([240, 80, 7])
    print("trace", mc.shape, "accept rate", mc[:, 2].mean(), "misfit range", mc[:, 0].min(), mc[:, 0].max())
    np.savez_compressed(os.path.join(HERE, "ref_driver.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_driver.npz"), os.path.getsize(os.path.join(HERE, "ref_driver.npz")))


if __name__ == "__main__":
    main()
