"""``PureGird`` (the reference's spelling, ``models.py:163-186``): a fixed 1-D model given as grid profiles - what
``Model1D.seisPropGrids()`` returns, e.g. a frozen posterior model - with the ``Model1D`` methods that lead into the hot path:
``seisPropGrids`` / ``seisPropLayers`` / ``value`` / ``moho`` / ``forward`` (``models.py:72-122``).  The profiles are cut into one
``PureGrid`` piece per group (``layers.py:96-106``), each restarted at depth 0 and stacked below the previous one - so a
profile without doubled points at its group boundaries closes up by the gap there, exactly as in the reference.  ``forward``
goes through ``forward._calForward`` -> the HIP ``fast_surf`` drop-in (no CPU fallback)."""
from __future__ import annotations

import numpy as np


class PureGird:
    def __init__(self, inProfiles, info=None):
        z, vs, vp, rho, qs, qp, grps = inProfiles
        prof = {k: np.asarray(v, float) for k, v in zip(("z", "vs", "vp", "rho", "qs", "qp"), (z, vs, vp, rho, qs, qp))}
        grps = np.asarray(list(grps))
        self._layers = []
        for grp in list(dict.fromkeys(grps.tolist())):            # groups in order of first appearance
            I = grps == grp
            piece = {k: v[I] for k, v in prof.items()}
            piece["z"] = piece["z"] - piece["z"][0]
            self._layers.append((grp, piece))
        self.info = info if info is not None else {}              # (the reference fails on info=None at the first .get)

    @property
    def layers(self):
        return self._layers

    def seisPropGrids(self, refLayer=False, hLowerLimit=0.01):
        if refLayer:
            raise NotImplementedError("a PureGird carries its own bottom; build the profiles with the reference mantle in them")
        z0 = -max(self.info.get("topo", 0), 0)                     # models.py:74
        cols = {k: [] for k in ("z", "vs", "vp", "rho", "qs", "qp")}
        grp = []
        for g, piece in self._layers:
            if piece["z"][-1] - piece["z"][0] < hLowerLimit:       # models.py:80-81
                continue
            cols["z"].append(piece["z"] + z0)
            for k in ("vs", "vp", "rho", "qs", "qp"):
                cols[k].append(piece[k])
            grp += [g] * len(piece["z"])
            z0 = cols["z"][-1][-1]
        out = [np.concatenate(cols[k]) for k in ("z", "vs", "vp", "rho", "qs", "qp")]
        return (*out, grp)

    def seisPropLayers(self, refLayer=False):
        z, vs, vp, rho, qs, qp, grp = self.seisPropGrids(refLayer)
        h = np.diff(z)
        mid = lambda a: (a[1:] + a[:-1]) / 2
        keep = h > 0.01                                            # models.py:102
        return (h[keep], mid(vs)[keep], mid(vp)[keep], mid(rho)[keep], mid(qs)[keep], mid(qp)[keep],
                list(np.array(grp[:-1])[keep]))

    def value(self, zdeps, type="vs"):
        if type != "vs":
            raise ValueError("Error: only support vs, others to be added...")
        z, vs, *_ = self.seisPropGrids()
        return np.interp(zdeps, z, vs, left=np.nan, right=np.nan)

    def moho(self):
        z, *_, grp = self.seisPropGrids()
        return z[grp.index("mantle")]

    def forward(self, periods=(5, 10, 20, 40, 60, 80)):
        """Rayleigh phase velocities at ``periods`` or None (models.py:115-122), through the HIP drop-in."""
        from .forward import _calForward
        return _calForward(np.array(self.seisPropLayers()[:-1]), wavetype="Ray", periods=list(periods))

    def copy(self):
        from copy import deepcopy
        return deepcopy(self)
