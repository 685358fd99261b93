"""Batched parameters -> layer stack (reference layer L3: ``Model1D.seisPropGrids/Layers``
``models.py:72-102`` and the layer classes of ``layers.py:139-284``), evaluated for B parameter
vectors at once with torch ops on the device, so that the Metropolis driver never leaves the GPU.

A *setting* is the reference's own dict (``models.py:42-51``; example ``point.py:373-391``):

    {'Sediment': {'H': 2., 'Vs': [[1.5,'abs',.5,.05], [2.2,'abs',.5,.05]]},
     'Crust':    {'H': [35.,'abs',10.,1.], 'Vs': [[3.4,'abs',.3,.02], ...]},
     'Mantle':   {'H': 160., 'Vs': [[4.4,'abs',.4,.02], ...]},
     'Info':     {'modelType': 'MCInv', 'refLayer': True}}

Random-walk entries are spelled ``[v, vmin, vmax, step]`` or ``[v, 'abs'|'abs_pos'|'rel'|'rel_pos',
width, step]`` (``layers.py:583-598``); everything else is a constant.  Parameter order = the
reference's ``MCinv._brownians()`` order (``models.py:240-253``): layers in dict order, keys in dict
order, list elements in order.

Supported layer types (rules cited per class): Sediment, Crust, Mantle/OceanMantle, OceanWater,
OceanSediment, OceanCrust, OceanSedimentCascadia, the thermal OceanMantleHybrid (SURVEY.md 8f-4,
``layers.py:297-363`` over ``thermseis.py``), and the ReferenceMantle appended when ``Info.refLayer``
is set (``models.py:116,153-154``).  The Crust ``Gauss`` option (``layers.py:176-183``) adds
``A exp(-(z - mu)^2 / (2 sigma^2))`` to the B-spline profile: the reference takes the function from the un-vendored
``Triforce.mathPlus.gaussFun(A, mu, sig, z)``, so the standard three-parameter form is assumed (pinned against the
reference code path with that form standing in, ``tests/golden/make_golden_driver.py``).

Per-point local information (``Point(setting, localInfo)``, ``point.py:8-14``; ``Model1D._loadLocalInfo``,
``models.py:54-59``): a grid run has one ``Info.topo`` / ``lithoAge`` / ``period`` - and possibly one fixed thickness
or velocity - PER POINT.  ``Model1DBatch(setting, local_keys=[...])`` declares which constants are per row
(``'topo'``, ``'lithoAge'``, ``'period'``, or ``'<Layer>.<key>'`` / ``'<Layer>.Vs[i]'`` for a layer constant);
``set_local_info(table[n_rows, K])`` puts the table on the device and every entry point takes ``rows`` (index of each
parameter vector's table row).  The values travel as extra constant columns behind the random-walk parameters - no
per-point Python objects, and the HIP kernels read them through the same slot indices.
"""
from __future__ import annotations

import numpy as np

from .brownian import ParamSpec

H_LOWER = 0.01            # models.py:72 hLowerLimit and models.py:102 h>0.01


def _is_brownian_entry(v):
    if isinstance(v, (list, tuple)):
        if len(v) >= 2 and isinstance(v[1], str) and v[1] in ("abs", "abs_pos", "rel", "rel_pos"):
            return True
        if len(v) == 4 and not isinstance(v[1], str):
            try:
                float(v[1]); return True
            except (TypeError, ValueError):
                return False
    return False


def _is_fixed_entry(v):
    return isinstance(v, (list, tuple)) and len(v) >= 2 and isinstance(v[1], str) and v[1] in ("fixed", "total")


def bspline_basis(npts, n_basis, deg=None, alpha=2.0):
    """Basis matrix [n_basis, npts] on z = linspace(0, 1, npts); same knot vector and recursion as
    the reference's ``BsplBasis`` (``layers.py:4-39``).  It does not depend on the layer thickness:
    the reference scales its knots with ``z[-1]-z[0]``."""
    z = np.linspace(0.0, 1.0, npts)
    if n_basis == 1:
        return np.ones((1, npts))
    if n_basis == 2:
        return np.stack([np.linspace(1, 0, npts), np.linspace(0, 1, npts)])
    eps = np.finfo(float).eps
    order = (3 + (n_basis >= 4)) if deg is None else int(deg)
    n = n_basis
    x = np.zeros(n + order)
    x[: order - 1] = -eps
    x[order - 1] = 0.0
    x[order:n] = np.power(alpha, np.arange(n - order)) * (alpha - 1) / (np.power(alpha, n - order + 1) - 1)
    x[n] = 1.0
    x[n + 1:] = 1.0 + eps
    ncol = len(x) - 1
    # order-1 (piecewise constant) functions, then raise the order one at a time (Cox - de Boor)
    cur = ((z[:, None] >= x[None, :-1]) & (z[:, None] < x[None, 1:])).astype(float)
    for k in range(1, order):
        nxt = cur.copy()                                   # columns not recomputed keep their values
        for i in range(ncol - k):
            col = np.zeros(npts)
            d1 = x[i + k] - x[i]
            d2 = x[i + k + 1] - x[i + 1]
            if d1 != 0:
                col += cur[:, i] * (z - x[i]) / d1
            if d2 != 0:
                col += cur[:, i + 1] * (x[i + k + 1] - z) / d2
            nxt[:, i] = col
        cur = nxt
    return cur[:, :n].T.copy()


def _n_fine_mantle(H):        # Crust / OceanMantle._nFineLayers, layers.py:156-168, 245-257
    import torch
    N = torch.full_like(H, 5, dtype=torch.int64)
    N = torch.where(H > 10, torch.full_like(N, 10), N)
    N = torch.where(H > 20, torch.full_like(N, 15), N)
    N = torch.where(H > 60, torch.full_like(N, 30), N)
    N = torch.where(H >= 150, torch.full_like(N, 60), N)
    return N


class _Slot:
    """A parameter value that is either a constant, or column ``idx`` of the parameter block (random-walk parameters
    first, then the per-row local constants: ``aux`` = index among those, resolved to ``idx`` once N is known)."""
    __slots__ = ("const", "idx", "aux")

    def __init__(self, const=None, idx=None, aux=None):
        self.const, self.idx, self.aux = const, idx, aux

    def get(self, params):
        import torch
        if self.idx is not None:
            return params[:, self.idx]
        return torch.full((params.shape[0],), float(self.const), dtype=params.dtype, device=params.device)


class Model1DBatch:
    LAYER_TYPES = {"Sediment": "sed", "Crust": "crust", "Mantle": "mantle", "OceanMantle": "mantle",
                   "OceanWater": "water", "OceanSediment": "osed", "OceanCrust": "ocrust",
                   "OceanSedimentCascadia": "osedc", "OceanMantleHybrid": "hybrid",
                   # the LayerName keys Model1D.toYML() writes (models.py:60-70) - what the reference stores as `setting` in
                   # its {pid}.npz files; its own layerClassDict does not know the two land names (layers.py:553-570)
                   "LandSediment": "sed", "LandCrust": "crust"}
    LAYER_NAME = {"sed": "LandSediment", "crust": "LandCrust", "mantle": "OceanMantle", "water": "OceanWater",
                  "osed": "OceanSediment", "ocrust": "OceanCrust", "osedc": "OceanSedimentCascadia",
                  "hybrid": "OceanMantleHybrid"}              # prop['LayerName'] of the reference's layer classes
    GROUP = {"sed": "sediment", "osed": "sediment", "osedc": "sediment", "crust": "crust", "ocrust": "crust",
             "mantle": "mantle", "hybrid": "mantle", "water": "water"}

    def __init__(self, setting, device="cpu", local_keys=None):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        if not isinstance(setting, dict):                  # a YAML file, as buildModel1D accepts (models.py:689-694)
            import yaml
            with open(setting, "r") as f:
                setting = yaml.load(f, Loader=yaml.FullLoader)
        self.setting = setting
        self.info = dict(setting.get("Info", {}))
        entries, names = [], []
        local_keys = list(local_keys or [])
        if len(set(local_keys)) != len(local_keys):
            raise ValueError("local_keys has duplicates")
        # columns of the local-information table = local_keys, in the caller's order
        self.aux_names, aux_slots, self._aux_default = list(local_keys), [], [None] * len(local_keys)
        self._aux = None                                   # [n_rows, K] float64 on the device (set_local_info)

        def slot(v, name):
            if name in local_keys:                         # a per-row constant (the setting's value = default)
                if _is_brownian_entry(v):
                    raise ValueError(f"{name}: a random-walk entry cannot be a per-point constant")
                val = float(v[0]) if _is_fixed_entry(v) else float(v)
                k = local_keys.index(name)
                if self._aux_default[k] is not None:
                    raise ValueError(f"{name} names more than one constant of the setting")
                sl = _Slot(const=val, aux=k)
                aux_slots.append(sl); self._aux_default[k] = val
                return sl
            if _is_brownian_entry(v):
                entries.append(list(v)); names.append(name)
                return _Slot(idx=len(entries) - 1)
            if _is_fixed_entry(v):
                return _Slot(const=float(v[0]))
            return _Slot(const=float(v))

        self.layers = []
        for key, parm in setting.items():
            if key == "Info":
                continue
            if key not in self.LAYER_TYPES:
                raise ValueError(f"layer type {key!r} is not supported by Model1DBatch")
            kind = self.LAYER_TYPES[key]
            lay = {"kind": kind, "name": key}
            for k, v in parm.items():                      # dict order = _brownians() order
                if k in ("H", "BottomDepth"):
                    lay["Hkey"] = k
                    lay["H"] = slot(v, f"{key}.{k}")
                elif k == "Vs":
                    if isinstance(v, (list, tuple)) and not _is_brownian_entry(v) and not _is_fixed_entry(v):
                        lay["Vs"] = [slot(e, f"{key}.Vs[{i}]") for i, e in enumerate(v)]
                    else:
                        lay["Vs"] = [slot(v, f"{key}.Vs")]
                        lay["Vs_scalar"] = True
                elif k == "deg":
                    # only the mantle's Vs profile reads 'deg' (OceanMantle._calVs, layers.py:258-260); Crust._calVs
                    # calls _bspl(z, nBasis) without it (layers.py:169-172)
                    if kind == "mantle":
                        lay["deg"] = int(v)
                elif k in ("ThermAge", "Tp") and kind == "hybrid":
                    lay[k] = slot(v, f"{key}.{k}")
                elif k == "Conversion" and kind == "hybrid":
                    if v not in ("Ritzwoller", "Yamauchi"):
                        raise ValueError(f"Invalid convertion model: {v}")       # layers.py:338
                    lay[k] = v
                elif k == "Gauss" and kind == "crust":
                    if v is not False and v is not None:   # layers.py:176: parm.get('Gauss', False) is False -> plain B-spline
                        if not (isinstance(v, (list, tuple)) and len(v) == 3):
                            raise ValueError("Crust 'Gauss' must be [A, mu, sigma]")
                        lay["Gauss"] = [slot(e, f"{key}.Gauss[{i}]") for i, e in enumerate(v)]
                elif _is_brownian_entry(v) or (isinstance(v, (list, tuple)) and any(_is_brownian_entry(e) for e in v)):
                    # the reference's _brownians() would count it (models.py:240-253): silently dropping it would
                    # shift every later mcTrack column against what PostPoint._loadMC expects
                    raise ValueError(f"{key}.{k}: random-walk entry under a key Model1DBatch does not implement")
            if kind in ("water", "osedc"):
                lay["Vs"] = []
            if kind == "hybrid":
                if "ThermAge" not in lay:
                    raise KeyError("OceanMantleHybrid needs 'ThermAge'")
                lay["deg"] = None                              # layers.py:341 calls _bspl(z, nBasis) only
                groups = [self.GROUP[l["kind"]] == "crust" for l in self.layers]
                runs = sum(1 for i, g in enumerate(groups) if g and (i == 0 or not groups[i - 1]))
                if runs != 1:                                  # getCrustH, layers.py:304-310
                    raise ValueError("OceanMantleHybrid: exactly one contiguous crust group must lie above it")
            self.layers.append(lay)
        # Info constants a grid run has per point (models.py:74 topo; layers.py:350-363 lithoAge, period)
        self._topo = slot(self.info.get("topo", 0.0), "topo")
        self._litho = slot(self.info.get("lithoAge", 0.0) if self.info.get("lithoAge", None) is not None else 0.0, "lithoAge")
        self._period = slot(self.info.get("period", 1), "period")
        unknown = [k for k, d in zip(local_keys, self._aux_default) if d is None]
        if unknown:
            raise ValueError(f"local_keys {unknown} name no constant of this setting")
        if "lithoAge" in self.aux_names and not self.info.get("lithoAgeQ", False):
            raise ValueError("per-point lithoAge only acts with Info.lithoAgeQ (layers.py:353)")
        self.spec = ParamSpec.from_entries(entries, names)
        for sl in aux_slots:                               # behind the random-walk parameters
            sl.idx = self.spec.n + sl.aux
        self._basis = {}
        self._static_sig = self._find_static_signature()

    def to_yml(self, params=None):
        """``Model1D.toYML()`` (models.py:60-70) of the model with the random-walk values ``params`` (default: the
        setting's own): {LayerName: parm with every random-walk entry as [v, vmin, vmax, step]} + ``Info`` - the form the
        reference stores as ``setting`` in its ``{pid}.npz`` files; ``Model1DBatch`` reads it back (the LayerName keys
        included)."""
        import copy
        v = np.asarray(self.spec.v0 if params is None else params, float).ravel()
        if v.size != self.spec.n:
            raise ValueError(f"{v.size} values for {self.spec.n} random-walk parameters")
        it = iter(range(self.spec.n))

        def conv(e):
            if _is_brownian_entry(e):
                i = next(it)
                return [float(v[i]), float(self.spec.vmin[i]), float(self.spec.vmax[i]), float(self.spec.step[i])]
            if _is_fixed_entry(e):
                return e[0]                                # buildSeisLayer turns 'fixed' / 'total' entries into plain values
            if isinstance(e, (list, tuple)):
                return [conv(x) for x in e]
            if isinstance(e, dict):
                return {kk: conv(x) for kk, x in e.items()}
            return copy.deepcopy(e)

        out = {}
        for lay, (key, parm) in zip(self.layers, ((kk, p) for kk, p in self.setting.items() if kk != "Info")):
            out[self.LAYER_NAME[lay["kind"]]] = {kk: conv(x) for kk, x in parm.items()}
            if lay["kind"] == "water":
                out[self.LAYER_NAME["water"]]["Vs"] = 0    # OceanWater.__init__ sets parm['Vs'] = 0 (layers.py:195)
        assert next(it, None) is None
        out["Info"] = copy.deepcopy(self.info)
        return out

    # ------------------------------------------------------------------ per-point local information
    @property
    def n_aux(self):
        return len(self.aux_names)

    def set_local_info(self, table):
        """``table`` [n_rows, K]: one row per point, columns in the order of ``aux_names`` (= ``local_keys``).  Kept on
        the device; the fine-layer structure is re-examined over the prior box x the table's range."""
        torch = self.torch
        t = torch.as_tensor(np.asarray(table, dtype=np.float64), dtype=torch.float64, device=self.device)
        if t.ndim != 2 or t.shape[1] != self.n_aux:
            raise ValueError(f"local info must be [n_rows, {self.n_aux}] with columns {self.aux_names}")
        self._aux = t.contiguous()
        self._static_sig = self._find_static_signature()
        self._native_desc = None
        return self

    def setting_for_row(self, values):
        """The setting dict of ONE point: ``self.setting`` with the per-point constants ``values`` (order of
        ``aux_names``) written back - what ``Point(setting, localInfo)`` would hold for it."""
        import copy
        import re
        st = copy.deepcopy(self.setting)
        for name, v in zip(self.aux_names, values):
            v = float(v)
            if "." not in name:
                st.setdefault("Info", {})[name] = v
                continue
            layer, key = name.split(".", 1)
            m = re.match(r"(\w+)\[(\d+)\]$", key)
            if m:
                st[layer][m.group(1)][int(m.group(2))] = v
            else:
                st[layer][key] = v
        return st

    def _full(self, params, rows=None):
        """Parameter block + this row's local constants [B, N + K]."""
        torch = self.torch
        params = params.to(torch.float64)
        if self.n_aux == 0:
            return params
        if self._aux is None:
            raise ValueError(f"this model has per-point constants {self.aux_names}: call set_local_info(table) first")
        if rows is None:
            if params.shape[0] == 1 and self._aux.shape[0] == 1:
                aux = self._aux[:1]
            elif params.shape[0] == 1:
                # (r03 took row 0 silently: a single-model call on a per-point model got point 0's topo / lithoAge / period)
                raise ValueError(f"one parameter vector against {self._aux.shape[0]} rows of local info: pass rows=[i]")
            elif params.shape[0] != self._aux.shape[0]:
                raise ValueError(f"{params.shape[0]} parameter vectors against {self._aux.shape[0]} rows of local info: pass rows=")
            else:
                aux = self._aux
        else:
            aux = self._aux[torch.as_tensor(rows, device=self.device, dtype=torch.int64)]
        return torch.cat([params, aux.to(params.device)], dim=1)

    def _find_static_signature(self):
        """Fine-layer counts are piecewise constant in the layer thickness.  If no thickness can cross
        a threshold anywhere inside the prior box, the layer structure is the same for every draw and
        the per-call signature search (a host synchronisation) is skipped."""
        import torch
        lo = torch.as_tensor(self.spec.vmin, dtype=torch.float64)[None, :]
        hi = torch.as_tensor(self.spec.vmax, dtype=torch.float64)[None, :]
        if self.n_aux:
            if self._aux is None:                          # no table yet: the setting's own values
                a_lo = a_hi = torch.as_tensor(self._aux_default, dtype=torch.float64)[None, :]
            else:
                a_lo, a_hi = self._aux.min(dim=0).values.cpu()[None, :], self._aux.max(dim=0).values.cpu()[None, :]
            lo, hi = torch.cat([lo, a_lo], dim=1), torch.cat([hi, a_hi], dim=1)
        # stack top z0 = -max(topo, 0) (models.py:74): deepest for the largest topo
        zlo = -self._topo.get(hi).clamp(min=0.0)
        zhi = -self._topo.get(lo).clamp(min=0.0)
        sig = []
        for lay in self.layers:
            h_lo, h_hi = lay["H"].get(lo), lay["H"].get(hi)
            if lay.get("Hkey") == "BottomDepth":
                h_lo, h_hi = h_lo - zhi, h_hi - zlo
            n_lo, n_hi = self._n_fine_from_H(lay, h_lo), self._n_fine_from_H(lay, h_hi)
            if int(n_lo) != int(n_hi) or float(h_lo) < H_LOWER * 4:      # also keeps every h > 0.01
                return None
            if lay["kind"] not in ("sed", "osed", "osedc", "water") and float(h_lo) / max(int(n_lo), 1) < H_LOWER * 2:
                return None
            sig.append(int(n_lo))
            zlo, zhi = zlo + h_lo, zhi + h_hi
        return sig

    def _n_fine_from_H(self, lay, H):
        torch = self.torch
        kind = lay["kind"]
        if kind in ("sed", "osed", "osedc", "water"):
            N = torch.ones_like(H, dtype=torch.int64)
        elif kind in ("crust", "mantle", "hybrid"):
            N = _n_fine_mantle(H)
        else:
            N = torch.clamp(torch.round(H / 2).to(torch.int64), 2, 10)
        return torch.where(H < H_LOWER, torch.full_like(N, -1), N)

    # ------------------------------------------------------------------ helpers
    def _bspl(self, N, n_basis, deg):
        key = (int(N), int(n_basis), deg)
        if key not in self._basis:
            self._basis[key] = self.torch.as_tensor(bspline_basis(int(N) + 1, int(n_basis), deg),
                                                    dtype=self.torch.float64, device=self.device)
        return self._basis[key]

    def _hybrid_grid(self, lay, params, z_bottom, crust_h, N, H, t):
        """OceanMantleHybrid (layers.py:297-363): thermal Vs of a half-space cooling model of age
        ``ThermAge`` above the depth where melting starts, thermal Vs + B-spline perturbation below 1.7 x
        that depth, a not-a-knot cubic spline through both parts in between; Qs from the pre-melting
        anelasticity model."""
        from . import thermseis as ts
        torch = self.torch
        z = t * H[:, None]
        age_p = lay["ThermAge"].get(params)
        age = age_p.clamp(min=1e-3)
        Tp = lay["Tp"].get(params)[:, None] if "Tp" in lay else 1325.0
        ther = ts.hscm(age, zdeps=crust_h[:, None] + z, Tp=Tp)
        if lay.get("Conversion", "Ritzwoller") == "Yamauchi":
            vs_th = ts.ruan(ther, period=1)[0]
        else:
            vs_th = ts.ritz_vs(ther)[0]
        z_melt = ts.melt_start(age, crust_h)
        coef = torch.stack([torch.zeros_like(age)] + [s.get(params) for s in lay["Vs"]], dim=1)
        y2 = coef @ self._bspl(N, coef.shape[1], None) + vs_th
        x_lo = z_melt[:, None]
        x_hi = ((z_melt + crust_h) * 1.7 - crust_h)[:, None]
        upper, lower = z < x_lo, z > x_hi
        vs = ts.cubic_spline_through(z, torch.where(upper, vs_th, y2), upper | lower)
        lay["zmelt_last"] = z_melt                      # the reference keeps it too (_debug_zMelt)
        lay["grid_last"] = None
        # _calOthers, layers.py:350-363
        if self.info.get("lithoAgeQ", False) and (self.info.get("lithoAge", None) is not None or self._litho.aux is not None):
            q_age = self._litho.get(params)
        else:
            q_age = age_p
        ther_q = ts.hscm(q_age.clamp(min=1e-3), zdeps=z_bottom[:, None] + z)
        period = self._period.get(params)[:, None] if self._period.aux is not None else float(self.info.get("period", 1))
        qs = ts.ruan(ther_q, period=period)[1].clamp(max=5000.0)
        lay["grid_last"] = (vs, qs)
        return z, vs, vs * 1.76, 3.4268 + (vs - 4.5) / 4.5, qs, 1400. * torch.ones_like(vs)

    def _layer_grid(self, lay, params, z_bottom, N, crust_h=None):
        """Grids (z[Bg,N+1] relative to the layer top, vs, vp, rho, qs, qp) for one layer type."""
        torch = self.torch
        Bg = params.shape[0]
        H = lay["H"].get(params)
        if lay.get("Hkey") == "BottomDepth":               # layers.py:119-124
            H = H - z_bottom
        t = torch.linspace(0.0, 1.0, N + 1, dtype=torch.float64, device=self.device)[None, :]
        z = t * H[:, None]
        kind = lay["kind"]
        ones = torch.ones((Bg, N + 1), dtype=torch.float64, device=self.device)
        if kind == "water":                                # layers.py:187-199
            return z, 0 * ones, 1.475 * ones, 1.027 * ones, 10000. * ones, 57822. * ones
        if kind == "hybrid":
            return self._hybrid_grid(lay, params, z_bottom, crust_h, N, H, t)
        if kind == "osedc":                                # OceanSedimentCascadia, layers.py:288-295
            vs = ((0.02 * H ** 2 + 1.27 * H + 0.29 * 0.1) / (H + 0.29))[:, None] * ones
            vp = vs * 1.23 + 1.28
            return z, vs, vp, 0.541 + 0.3601 * vp, 80. * ones, 160. * ones
        coef = torch.stack([s.get(params) for s in lay["Vs"]], dim=1)      # [Bg, nb]
        nb = coef.shape[1]
        if kind in ("sed", "ocrust"):                      # layers.py:145-149, 218-222
            vs = coef[:, :1] + (coef[:, 1:2] - coef[:, :1]) * t if nb == 2 else coef[:, :1] * ones
        elif kind == "osed":                               # layers.py:206-207
            vs = coef[:, :1] * ones
        else:                                              # B-spline layers, layers.py:169-172, 258-261
            vs = coef @ self._bspl(N, nb, lay.get("deg"))
            if kind == "crust" and "Gauss" in lay:             # layers.py:176-183: + gaussFun(A, mu, sig, z)
                ga, gm, gs = (sl.get(params)[:, None] for sl in lay["Gauss"])
                vs = vs + ga * torch.exp(-((z - gm) ** 2) / (2.0 * gs * gs))
        if kind == "sed":                                  # layers.py:150-155
            vp = vs * 2.0
            rho = 1.22679 + 1.53201 * vs - 0.83668 * vs * vs + 0.20673 * vs ** 3 - 0.01656 * vs ** 4
            qs, qp = 80. * ones, 160. * ones
        elif kind == "crust":                              # layers.py:179-184
            vp = vs * 1.80
            rho = 1.22679 + 1.53201 * vs - 0.83668 * vs * vs + 0.20673 * vs ** 3 - 0.01656 * vs ** 4
            qs, qp = 600. * ones, 1400. * ones
        elif kind == "mantle":                             # layers.py:262-267
            vp = vs * 1.76
            rho = 3.4268 + (vs - 4.5) / 4.5
            qs, qp = 150. * ones, 1400. * ones
        elif kind == "osed":                               # layers.py:208-213
            vp = vs * 1.23 + 1.28
            rho = 0.541 + 0.3601 * vp
            qs, qp = 80. * ones, 160. * ones
        elif kind == "ocrust":                             # layers.py:223-228
            vp = vs * 1.8
            rho = 0.541 + 0.3601 * vp
            qs, qp = 350. * ones, 1400. * ones
        else:
            raise AssertionError(kind)
        return z, vs, vp, rho, qs, qp

    def _n_fine(self, lay, params, z_bottom):
        torch = self.torch
        H = lay["H"].get(params)
        if lay.get("Hkey") == "BottomDepth":
            H = H - z_bottom
        kind = lay["kind"]
        if kind in ("sed", "osed", "osedc", "water"):
            N = torch.ones_like(H, dtype=torch.int64)
        elif kind in ("crust", "mantle", "hybrid"):
            N = _n_fine_mantle(H)
        else:                                              # OceanCrust, layers.py:216-217
            N = torch.clamp(torch.round(H / 2).to(torch.int64), 2, 10)
        return torch.where(H < H_LOWER, torch.full_like(N, -1), N), H     # -1: layer skipped (models.py:80-81)

    # ------------------------------------------------------------------ public
    def _grid_groups(self, params, ref_layer=None):
        """The grid points of every row of ``params`` ([B, N + K], local constants appended), grouped by fine-layer signature:
        yields (rows, (z, vs, vp, rho, qs, qp) float64 [n, G], group codes int [G]) - ``Model1D.seisPropGrids(refLayer)``
        (models.py:72-91): interface points doubled, layers thinner than hLowerLimit skipped.  Also returns whether the
        structure is static and the signatures (for the layer count of ``seis_prop_layers``)."""
        torch = self.torch
        B = params.shape[0]
        ref_layer = bool(self.info.get("refLayer", False)) if ref_layer is None else bool(ref_layer)
        z_start = -self._topo.get(params).clamp(min=0.0)   # models.py:74, per row
        # pass 1: fine-layer counts (they depend on the thicknesses, hence on the parameters)
        static = self._static_sig is not None
        if static:
            uniq = torch.as_tensor([self._static_sig], dtype=torch.int64)
            inv = None
        else:
            zb = z_start.clone()
            sig = []
            for lay in self.layers:
                N, H = self._n_fine(lay, params, zb)
                sig.append(N)
                zb = torch.where(N >= 0, zb + H, zb)
            sig = torch.stack(sig, dim=1)                      # [B, nlayers]
            uniq, inv = torch.unique(sig, dim=0, return_inverse=True)
            uniq = uniq.cpu()
        groups = []
        for g in range(uniq.shape[0]):
            rows = slice(None) if static else (inv == g).nonzero(as_tuple=True)[0]
            p = params[rows]
            zbot = z_start[rows].clone()
            cols = [[] for _ in range(6)]
            codes = []
            crust_h = torch.zeros_like(zbot)
            for li, lay in enumerate(self.layers):
                N = int(uniq[g, li].item())
                if N < 0:
                    continue
                z, vs, vp, rho, qs, qp = self._layer_grid(lay, p, zbot, N, crust_h)
                for c_, a in zip(cols, (z + zbot[:, None], vs, vp, rho, qs, qp)):
                    c_.append(a)
                codes += [self.GROUP_CODE[self.GROUP[lay["kind"]]]] * z.shape[1]
                zbot = zbot + z[:, -1]
                if self.GROUP[lay["kind"]] == "crust":
                    crust_h = crust_h + z[:, -1]
            if ref_layer:                                  # ReferenceMantle, layers.py:267-284
                vs0, vp0, rho0, qs0, qp0 = (cols[i][-1][:, -1:] for i in range(1, 6))
                t = torch.linspace(0.0, 1.0, 21, dtype=torch.float64, device=self.device)[None, :]
                zr = t * 300.0
                vs = vs0 + zr * (0.35 / 200)
                vp = vp0 + (vs * 1.76 - vs[:, :1] * 1.76)
                rho = rho0 + ((3.4268 + (vs - 4.5) / 4.5) - (3.4268 + (vs[:, :1] - 4.5) / 4.5))
                one = torch.ones_like(zr)
                for c_, a in zip(cols, (zr + zbot[:, None], vs, vp, rho, qs0 * one, qp0 * one)):
                    c_.append(a)
                codes += [self.GROUP_CODE["mantle"]] * 21
            groups.append((rows, tuple(torch.cat(c_, dim=1) for c_ in cols), codes))
        return groups, static, uniq, ref_layer

    GROUP_NAMES = ("water", "sediment", "crust", "mantle")
    GROUP_CODE = {n: i for i, n in enumerate(GROUP_NAMES)}

    def seis_prop_grids(self, params, rows=None, ref_layer=None):
        """``Model1D.seisPropGrids(refLayer)`` (models.py:72-91; ``ref_layer=None``: Info.refLayer) for every row of ``params``:
        (z, vs, vp, rho, qs, qp) float64 [B, Gmax] padded with zeros - the grid points of every layer from the surface
        down, interface points doubled as in the reference -, grp int64 [B, Gmax] (index into ``GROUP_NAMES``, -1 in the
        padding) and ngrid[B]."""
        torch = self.torch
        params = self._full(params, rows)
        B = params.shape[0]
        groups, _, _, _ = self._grid_groups(params, ref_layer)
        Gmax = max(g[1][0].shape[1] for g in groups)
        out = [torch.zeros((B, Gmax), dtype=torch.float64, device=self.device) for _ in range(6)]
        grp = torch.full((B, Gmax), -1, dtype=torch.int64, device=self.device)
        ngrid = torch.zeros(B, dtype=torch.int32, device=self.device)
        for rws, arrs, codes in groups:
            n = arrs[0].shape[1]
            for o, a in zip(out, arrs):
                o[rws, :n] = a
            grp[rws, :n] = torch.as_tensor(codes, dtype=torch.int64, device=self.device)[None, :]
            ngrid[rws] = n
        return tuple(out), grp, ngrid

    def value(self, params, zdeps, rows=None, type="vs"):
        """``Model1D.value(zdeps)`` (models.py:104-108) for every row of ``params``: Vs at the depths ``zdeps``, linearly
        interpolated on ``seisPropGrids()`` - WITHOUT the reference mantle, as the reference calls it -, NaN outside.
        float64 numpy [B, len(zdeps)] (post-processing: numpy's own ``interp``, row by row, so that doubled interface
        points resolve exactly as in the reference)."""
        if type != "vs":
            raise ValueError("Error: only support vs, others to be added...")       # models.py:105-106
        (z, vs, *_), _, ngrid = self.seis_prop_grids(params, rows, ref_layer=False)
        z, vs, ngrid = z.cpu().numpy(), vs.cpu().numpy(), ngrid.cpu().numpy()
        zd = np.asarray(zdeps, float)
        return np.stack([np.interp(zd, z[i, :n], vs[i, :n], left=np.nan, right=np.nan) for i, n in enumerate(ngrid)])

    def moho(self, params, rows=None):
        """``Model1D.moho()`` (models.py:110-112): depth of the first mantle grid point, float64 numpy [B]."""
        (z, *_), grp, _ = self.seis_prop_grids(params, rows, ref_layer=False)
        first = (grp == self.GROUP_CODE["mantle"]).to(self.torch.int8).argmax(dim=1)
        if not bool((grp == self.GROUP_CODE["mantle"]).any(dim=1).all()):
            raise ValueError("'mantle' is not in list")                              # what list.index raises in the reference
        return z.gather(1, first[:, None]).squeeze(1).cpu().numpy()

    def seis_prop_layers(self, params, rows=None):
        """(h, vs, vp, rho, qs, qp) float64 [B, Lmax] padded with zeros, and nlay[B] -
        ``Model1D.seisPropLayers(refLayer=Info.refLayer)`` for every row of ``params`` (``rows``: the local-info row of
        each parameter vector, see ``set_local_info``)."""
        torch = self.torch
        params = self._full(params, rows)
        B = params.shape[0]
        groups, static, uniq, ref_layer = self._grid_groups(params)
        Lcap = int((uniq.clamp(min=0) + 1).sum(dim=1).max().item()) + (21 if ref_layer else 0)
        out = [torch.zeros((B, Lcap), dtype=torch.float64, device=self.device) for _ in range(6)]
        nlay = torch.zeros(B, dtype=torch.int32, device=self.device)
        for rws, (z, vs, vp, rho, qs, qp), _ in groups:
            h = z[:, 1:] - z[:, :-1]                       # models.py:95-101
            mids = [h] + [(a[:, 1:] + a[:, :-1]) / 2 for a in (vs, vp, rho, qs, qp)]
            keep = h > H_LOWER
            order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
            n = keep.sum(dim=1)
            L = h.shape[1]
            valid = torch.arange(L, device=self.device)[None, :] < n[:, None]
            for o, a in zip(out, mids):
                a = torch.gather(a, 1, order) * valid
                o[rws, :L] = a
            nlay[rws] = n.to(torch.int32)
        # static structure: the interface duplicates are the only rows ever dropped, their number is
        # known, so the width is known without asking the device
        Lmax = (Lcap - 1 - (len([n for n in self._static_sig if n >= 0]) - 1 + (1 if ref_layer else 0))
                if static else int(nlay.max().item()))
        return tuple(o[:, :Lmax] for o in out), nlay

    # ------------------------------------------------------------------ native (HIP) path
    KIND_CODE = {"sed": 0, "crust": 1, "mantle": 2, "water": 3, "osed": 4, "ocrust": 5, "hybrid": 6, "osedc": 7}

    def native_descriptor(self):
        """(idesc int32, fdesc float64, L) for csrc/surfdisp_layers.hip, or None when the layer
        structure is not static.  Cached on the model's device."""
        if self._static_sig is None or len(self.layers) > 10 or any("Gauss" in lay for lay in self.layers):
            return None                                    # (the Gaussian crust term is evaluated by the torch path)
        if self.n_aux and self._aux is None:
            return None
        hybrids = [i for i, lay in enumerate(self.layers) if lay["kind"] == "hybrid"]
        if len(hybrids) > 1 or (hybrids and self._static_sig[hybrids[0]] + 1 > 64):
            return None
        if getattr(self, "_native_desc", None) is not None:
            return self._native_desc
        torch = self.torch
        nin = len(self.layers)
        ref = bool(self.info.get("refLayer", False))
        lay_i, coef_i, lay_f, grid_f, tops = [], [], [], [], []
        g = 0
        for lay, N in zip(self.layers, self._static_sig):
            hs = lay["H"]
            slots = lay["Vs"]
            if lay["kind"] == "hybrid":                        # leading zero coefficient, layers.py:341
                slots = [_Slot(const=0.0)] + list(slots)
            if len(slots) > 8:
                return None
            begin, end = g, g + N + 1
            # int 6 of the FIRST layer: slot + 1 of a per-row Info.topo (0: the constant z_start of fdesc[0])
            topo_ref = (self._topo.idx + 1) if (not lay_i and self._topo.idx is not None) else 0
            lay_i += [self.KIND_CODE[lay["kind"]], hs.idx if hs.idx is not None else -1,
                      1 if lay.get("Hkey") == "BottomDepth" else 0, len(slots), begin, end, topo_ref, 0]
            coef_i += [(sl.idx if sl.idx is not None else -1) for sl in slots] + [-1] * (8 - len(slots))
            lay_f += [float(hs.const) if hs.const is not None else 0.0]
            lay_f += [(float(sl.const) if sl.const is not None else 0.0) for sl in slots] + [0.0] * (8 - len(slots))
            kind, nb = lay["kind"], len(slots)
            bas = None
            if kind in ("crust", "mantle", "hybrid"):
                bas = bspline_basis(N + 1, nb, lay.get("deg"))          # [nb, N+1]
            for q in range(N + 1):
                t = float(np.linspace(0.0, 1.0, N + 1)[q])
                if kind == "water":
                    row = []
                elif kind in ("crust", "mantle", "hybrid"):
                    row = [float(bas[k, q]) for k in range(nb)]
                elif kind == "osedc":
                    row = []
                elif nb == 2 and kind in ("sed", "ocrust"):
                    row = [1.0 - t, t]
                else:
                    row = [1.0]
                grid_f += [t] + row + [0.0] * (8 - len(row))
            tops += list(range(begin, end - 1))
            g = end
        ngrid = g
        if ref:
            tops += list(range(ngrid, ngrid + 20))
        L = len(tops)
        idesc = [nin, ngrid, L, 1 if ref else 0] + lay_i + coef_i + tops
        topo = float(self._topo.const if self._topo.const is not None else 0.0)
        fdesc = [-max(topo, 0.0)] + lay_f + grid_f
        self._native_thermal = bool(hybrids)
        if hybrids:                                        # descriptor tail, csrc/surfdisp_thermal.hip
            lh = hybrids[0]
            lay = self.layers[lh]
            age, tp = lay["ThermAge"], lay.get("Tp", _Slot(const=1325.0))
            q_const = bool(self.info.get("lithoAgeQ", False)) and (self.info.get("lithoAge", None) is not None
                                                                   or self._litho.aux is not None)
            crust_mask = sum(1 << i for i, l in enumerate(self.layers[:lh]) if self.GROUP[l["kind"]] == "crust")
            # ints 7, 8: slot + 1 of a per-row Info.lithoAge / Info.period (0: the constants of the fdesc tail)
            idesc += [lh, age.idx if age.idx is not None else -1, tp.idx if tp.idx is not None else -1,
                      1 if lay.get("Conversion", "Ritzwoller") == "Yamauchi" else 0, 1 if q_const else 0,
                      crust_mask, self._static_sig[lh] + 1,
                      (self._litho.idx + 1) if self._litho.idx is not None else 0,
                      (self._period.idx + 1) if self._period.idx is not None else 0, 0]
            fdesc += [float(age.const) if age.const is not None else 0.0,
                      float(tp.const) if tp.const is not None else 0.0,
                      float(self._litho.const) if q_const else 0.0, float(self._period.const)]
        self._native_desc = (torch.tensor(idesc, dtype=torch.int32, device=self.device),
                             torch.tensor(fdesc, dtype=torch.float64, device=self.device), L)
        return self._native_desc

    def prior_flags(self, rules):
        """int32 [L] flags of csrc/surfdisp_layers.hip::surfdisp_prior_kernel for a ``PriorRules`` (per OUTPUT layer of the
        native descriptor: bit 0 Vs increases across the layer, bit 1 ... and on to the next layer's top, bit 2 Vs does not
        drop to the next layer's top, bit 3 the Vs cap applies), or None when the model has no native descriptor / a thermal layer."""
        desc = self.native_descriptor()
        if desc is None or self._native_thermal:
            return None
        flags = []
        groups = [self.GROUP[l["kind"]] for l in self.layers]
        for l, (lay, N) in enumerate(zip(self.layers, self._static_sig)):
            mono = groups[l] in rules.monotone_groups
            for q in range(N):                                    # the N output layers of input layer l
                f = (1 if mono else 0) | (8 if rules.vs_max is not None else 0)
                if q == N - 1 and l + 1 < len(self.layers):
                    if groups[l + 1] == groups[l]:
                        f |= 2 if mono else 0
                    elif rules.positive_jumps:
                        f |= 4
                flags.append(f)
        flags += [0] * (desc[2] - len(flags))                   # (the ReferenceMantle's layers carry no rule)
        return self.torch.tensor(flags, dtype=self.torch.int32, device=self.device)

    def prior_good(self, params, rules, rows=None):
        """bool [B]: ``rules`` evaluated in torch on ``seis_prop_grids`` - the reference's own formulation (models.py:294-320:
        tests on the grid points' Vs and group names); what the device kernel is checked against."""
        torch = self.torch
        (z, vs, vp, rho, qs, qp), grp, ngrid = self.seis_prop_grids(params, rows=rows, ref_layer=False)
        if not bool((grp == grp[:1]).all()):
            raise NotImplementedError("prior_good: every row must have the same layer structure")
        nm = np.asarray([self.GROUP_NAMES[int(q)] for q in grp[0].tolist() if int(q) >= 0])
        vs = vs[:, :nm.size]
        ok = torch.ones(vs.shape[0], dtype=torch.bool, device=vs.device)
        eps = float(np.finfo(float).eps)
        if rules.positive_jumps:                                   # models.py:302-307
            for i in np.where(nm[1:] != nm[:-1])[0]:
                ok &= ~(vs[:, i + 1] < vs[:, i])
        if rules.vs_max is not None:                               # models.py:309-313
            ok &= ~(vs > rules.vs_max).any(dim=1)
        for g in rules.monotone_groups:                            # models.py:315-320 (monoIncrease, :8-9)
            sel = torch.as_tensor(np.where(nm == g)[0], device=vs.device)
            if sel.numel() > 1:
                v = vs[:, sel]
                ok &= (v[:, 1:] - v[:, :-1] >= eps).all(dim=1)
        return ok

    def to_model_native(self, params, rows=None):
        """``to_model`` through the HIP kernel surfdisp_layers_kernel (one launch, graph-capturable)."""
        import ctypes
        from . import _lib
        torch = self.torch
        desc = self.native_descriptor()
        if desc is None or params.device.type != "cuda":
            raise _lib.SurfdispError("native parameters->stack needs a static layer structure and a HIP device")
        idesc, fdesc, L = desc
        p = self._full(params, rows).contiguous()          # [C, N + K]: the kernels index the local constants like parameters
        C, N = p.shape
        model = torch.empty((C, 5, L), dtype=torch.float32, device=p.device)
        stream = torch.cuda.current_stream(p.device).cuda_stream
        with torch.cuda.device(p.device):
            if self._native_thermal:
                scratch = torch.empty((C, 64, 2), dtype=torch.float64, device=p.device)
                self._thermal_scratch = scratch                # (vs, qs) per grid point, kept for inspection
                rc = _lib.lib().surfdisp_params_to_model_thermal_device(
                    ctypes.c_void_p(stream), C, N, L, ctypes.c_void_p(p.data_ptr()),
                    ctypes.c_void_p(idesc.data_ptr()), ctypes.c_void_p(fdesc.data_ptr()),
                    ctypes.c_void_p(scratch.data_ptr()), scratch.numel() * 8, ctypes.c_void_p(model.data_ptr()))
            else:
                rc = _lib.lib().surfdisp_params_to_model_device(
                    ctypes.c_void_p(stream), C, N, L, ctypes.c_void_p(p.data_ptr()),
                    ctypes.c_void_p(idesc.data_ptr()), ctypes.c_void_p(fdesc.data_ptr()),
                    ctypes.c_void_p(model.data_ptr()))
        _lib.check(rc)
        return model, None

    def to_model(self, params, rows=None):
        """model float32 [B, 5, Lmax] rows (vp, vs, rho, h, 1/Qs) + nlay - what ``_calForward``
        hands to ``fast_surf`` (models.py:20-27).  On a HIP device with a static layer structure this
        is one kernel launch (``to_model_native``); otherwise the torch implementation below.  ``rows``: the local-info
        row of each parameter vector (models with per-point constants, ``set_local_info``)."""
        if params.device.type == "cuda" and self.native_descriptor() is not None:
            return self.to_model_native(params, rows)
        return self.to_model_torch(params, rows)

    def forward(self, params=None, periods=(5, 10, 20, 40, 60, 80), wavetype="Ray", rows=None):
        """``Model1D.forward(periods)`` (models.py:116-122) for every row of ``params`` (default: the
        setting's own values): phase velocities float32 [B, P] through the HIP solver, rows of zeros where
        the reference would return ``None``; status [B]."""
        import numpy as _np
        from . import _lib, forward as _fw
        torch = self.torch
        if self.device.type != "cuda":
            raise _lib.SurfdispError("Model1DBatch.forward needs a HIP device (no CPU fallback)")
        if params is None:
            params = torch.as_tensor(self.spec.v0[None, :], dtype=torch.float64, device=self.device)
        model, nlay = self.to_model(params, rows)
        per = torch.as_tensor(_np.asarray(periods, _np.float32), device=self.device)
        kind = {"Ray": _lib.KIND_RAYLEIGH, "Love": _lib.KIND_LOVE}[wavetype] | _lib.PHASE_ONLY
        plan = _fw.BatchPlan(model.shape[0], model.shape[2], per.numel(), device=self.device)
        c, _, st = plan.run(model.contiguous(), per, kind=kind, nlay=nlay)
        return c, st

    def to_model_torch(self, params, rows=None):
        torch = self.torch
        (h, vs, vp, rho, qs, qp), nlay = self.seis_prop_layers(params, rows)
        qsinv = torch.where(qs > 0, 1.0 / torch.where(qs > 0, qs, torch.ones_like(qs)), torch.zeros_like(qs))
        model = torch.stack([vp, vs, rho, h, qsinv], dim=1).to(torch.float32).contiguous()
        return model, nlay
