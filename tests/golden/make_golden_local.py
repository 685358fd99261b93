#!/usr/bin/env python3
"""Golden vectors for per-point local information and the Crust 'Gauss' option, from the imported reference Python
(development container only; see _refimport.py; same mechanism as make_golden_driver.py):

  every grid point builds ITS OWN model, ``buildModel1D(setting_i, localInfo_i)`` (``Point.__init__``, point.py:8-14;
  ``Model1D._loadLocalInfo`` merges localInfo into Info, models.py:54-59), and the fixture holds, per point, the
  parameter vectors of the initial model and of ``reset()`` draws with their ``seisPropLayers(refLayer)`` outputs:

  * ``hyb``   thermal oceanic model (settings_therm.HYBRID_STATIC): per-point ``topo``, ``lithoAge`` (Q age, lithoAgeQ),
              ``period`` (Q period) and a per-point fixed water depth ``OceanWater.H``;
  * ``ocean`` settings.OCEAN (mantle given by BottomDepth): per-point ``topo`` moves the stack's top (models.py:74);
  * ``gauss`` continental model whose crust carries ``Gauss = [A (random walk), mu (per point), sigma]``.

    python tests/golden/make_golden_local.py        -> tests/golden/ref_local.npz
"""
import copy
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
from settings_therm import HYBRID_STATIC             # noqa: E402
from settings_local import GAUSS, LOCAL_TABLES       # noqa: E402


def point_setting(setting, keys, row):
    """(setting_i, localInfo_i) of one point: Info-level keys go through localInfo as Point() passes them, layer
    constants are written into the point's own copy of the setting."""
    st, info = copy.deepcopy(setting), {}
    for k, v in zip(keys, row):
        if "." not in k:
            info[k] = float(v)
        else:
            layer, key = k.split(".", 1)
            if key.endswith("]"):
                name, i = key[:-1].split("[")
                st[layer][name][int(i)] = float(v)
            else:
                st[layer][key] = float(v)
    return st, info


def capture(setting, keys, table, ndraw, seed):
    random.seed(seed)
    ref = setting['Info'].get('refLayer', False)
    P, H, NL = [], [], []
    for row in table:
        st, info = point_setting(setting, keys, row)
        mod0 = buildModel1D(st, info)
        for m in [mod0] + [mod0.reset() for _ in range(ndraw - 1)]:
            P.append(m._brownians())
            out = m.seisPropLayers(refLayer=ref)[:-1]
            H.append(np.array(out)); NL.append(len(out[0]))
    Lmax = max(NL)
    arr = np.zeros((len(H), 6, Lmax))
    for i, a in enumerate(H):
        arr[i, :, :a.shape[1]] = a
    rows = np.repeat(np.arange(len(table)), ndraw)
    return dict(params=np.array(P), layers=arr, nlay=np.array(NL), rows=rows, table=np.asarray(table, float))


def main():
    out = {}
    for name, setting in (("hyb", HYBRID_STATIC), ("ocean", OCEAN), ("gauss", GAUSS)):
        keys, table = LOCAL_TABLES[name]
        d = capture(setting, keys, table, 3, seed=31)
        for k, v in d.items():
            out[f"{name}/{k}"] = v
        print(name, keys, "points", len(table), "layers", d["layers"].shape, "nlay", np.unique(d["nlay"]))
    np.savez_compressed(os.path.join(HERE, "ref_local.npz"), **out)
    print("wrote ref_local.npz")


if __name__ == "__main__":
    main()
