#!/usr/bin/env python3
"""Captures golden vectors for SURVEY.md row 8f-4 from the imported reference Python (development
container only; see _refimport.py):

  * ``HSCM(age, zdeps, Tp)`` fields T, P, rho (ThermSeis.py:56-101);
  * ``OceanSeisRitz`` vs for the three RhoType options, ``OceanSeisRuan`` / ``OceanSeisYaTa`` vs, qs,
    vs_unrelaxed, ``OceanSeisBass/Stix/PM13/YaTa_unrelaxed`` vs, ``behn2009Shear``;
  * ``Model1D.seisPropLayers(refLayer)`` of an oceanic setting with ``OceanSedimentCascadia`` and
    ``OceanMantleHybrid`` (layers.py:288-363), both conversions, for parameter vectors drawn by the
    reference's own ``reset()``, plus ``_debug_zMelt`` and the forward prediction.

    python tests/golden/make_golden_therm.py        -> tests/golden/ref_therm.npz
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.install()
from pySurfInv import ThermSeis as TS                # noqa: E402
from pySurfInv.models import buildModel1D            # noqa: E402
from settings_therm import HYBRID_RITZ, HYBRID_YAMA, PERIODS   # noqa: E402

AGES = [1e-3, 0.05, 0.6, 4.0, 20.0, 80.0, 170.0]


def capture_hybrid(setting, nsamp, seed):
    random.seed(seed)
    mod0 = buildModel1D(setting)
    ref = setting['Info'].get('refLayer', False)
    P, H, NL, C, ZM = [], [], [], [], []
    mods = [mod0] + [mod0.reset() for _ in range(nsamp - 1)]
    for m in mods:
        P.append(m._brownians())
        out = m.seisPropLayers(refLayer=ref)[:-1]
        H.append(np.array(out)); NL.append(len(out[0]))
        ZM.append(m._layers[-1]._debug_zMelt)
    Lmax = max(NL)
    arr = np.zeros((nsamp, 6, Lmax))
    for i, a in enumerate(H):
        arr[i, :, :a.shape[1]] = a
    for m in mods[:6]:
        c = m.forward(PERIODS)
        C.append(np.zeros(len(PERIODS)) if c is None else np.array(c))
    return dict(params=np.array(P), layers=arr, nlay=np.array(NL), c=np.array(C), zmelt=np.array(ZM))


def main():
    out = {"ages": np.array(AGES)}
    zc = 7.0 + np.linspace(0, 186.5, 61)
    for tag, kw in (("default", {}), ("custom", dict(zdeps=zc, Tp=1350))):
        T, P, R = [], [], []
        rz = {k: [] for k in ("raw", "corrected", "from_thermal")}
        ru = {k: [] for k in ("vs", "qs", "vsu")}
        ru10 = {k: [] for k in ("vs", "qs")}
        ya = {k: [] for k in ("Takei2017", "Hirschmann2009", "Ruan2018")}
        yaq = {k: [] for k in ya}
        oth = {k: [] for k in ("bass", "stix", "pm13", "yata_unrelaxed")}
        for age in AGES:
            th = TS.HSCM(age, **kw)
            T.append(th.T); P.append(th.P); R.append(th.rho)
            for k in rz:
                rz[k].append(TS.OceanSeisRitz(th, RhoType=k).vs)
            m = TS.OceanSeisRuan(th, period=1)
            ru["vs"].append(m.vs); ru["qs"].append(m.qs); ru["vsu"].append(m.vs_unrelaxed)
            m = TS.OceanSeisRuan(th, period=10)
            ru10["vs"].append(m.vs); ru10["qs"].append(m.qs)
            for k in ya:
                m = TS.OceanSeisYaTa(th, Tm=k, period=50)
                ya[k].append(m.vs); yaq[k].append(m.qs)
            oth["bass"].append(TS.OceanSeisBass(th).vs)
            oth["stix"].append(TS.OceanSeisStix(th).vs)
            oth["pm13"].append(TS.OceanSeisPM13(th, period=1).vs)
            oth["yata_unrelaxed"].append(TS.OceanSeisYaTa_unrelaxed(th).vs)
        out[f"hscm/{tag}/zdeps"] = np.array(th.zdeps)
        out[f"hscm/{tag}/T"] = np.array(T); out[f"hscm/{tag}/P"] = np.array(P); out[f"hscm/{tag}/rho"] = np.array(R)
        for k, v in rz.items():
            out[f"ritz/{tag}/{k}"] = np.array(v)
        for k, v in ru.items():
            out[f"ruan1/{tag}/{k}"] = np.array(v)
        for k, v in ru10.items():
            out[f"ruan10/{tag}/{k}"] = np.array(v)
        for k in ya:
            out[f"yata50/{tag}/{k}/vs"] = np.array(ya[k]); out[f"yata50/{tag}/{k}/qs"] = np.array(yaq[k])
        for k, v in oth.items():
            out[f"other/{tag}/{k}"] = np.array(v)
    Tc = np.array([900., 1100., 1300.]); Pg = np.array([1.0, 2.0, 4.0])
    q, sf = TS.behn2009Shear(1.0, 1e-3, Tc, Pg, 100)
    out["behn/T"], out["behn/P"], out["behn/Qinv"], out["behn/shear"] = Tc, Pg, q, sf
    for name, setting in (("hyb_ritz", HYBRID_RITZ), ("hyb_yama", HYBRID_YAMA)):
        d = capture_hybrid(setting, 24, seed=23)
        for k, v in d.items():
            out[f"{name}/{k}"] = v
        print(name, "layers", d["layers"].shape, "nlay", np.unique(d["nlay"]), "npar", d["params"].shape[1],
              "zmelt", d["zmelt"].min(), d["zmelt"].max(), "c", d["c"][0][:3])
    np.savez_compressed(os.path.join(HERE, "ref_therm.npz"), **out)
    print("wrote ref_therm.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
