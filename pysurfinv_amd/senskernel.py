"""Finite-difference sensitivity kernels: the batched counterpart of ``SensKernelPert``
(``/root/reference/senskernel.py:129-158``).

The reference calls ``fast_surf`` 2L(+L) times, one perturbed stack at a time: layer i's Vs (or Vp)
scaled by 0.999 and 1.001, ``kernel[:, i] = (v(1.001) - v(0.999)) / 0.2 / H[i]`` (``:147-150``).
Here the 2L perturbed stacks (plus the unperturbed one) are ONE batch through the HIP solver.

Differences from the reference, on purpose:
* ``wtype='L'``: the reference reads ``cr0`` even for Love (``senskernel.py:188-192``, SURVEY.md
  section 4 defect 7) so its Love kernels are ``None``; here Love uses the Love phase velocities.
* group-velocity kernels (``ytype='grv'``) come for free from the same batch.
"""
from __future__ import annotations

import numpy as np

from . import forward as _forward

GROUP_RULES = {            # sensModel._convert, senskernel.py:104-124
    "water": lambda vs: (np.full_like(vs, 1.475), np.full_like(vs, 1.027), np.full_like(vs, 10000.)),
    "sediment": lambda vs: (vs * 1.23 + 1.28, 0.541 + 0.3601 * (vs * 1.23 + 1.28), np.full_like(vs, 80.)),
    "crust": lambda vs: (vs * 1.8, 0.541 + 0.3601 * (vs * 1.8), np.full_like(vs, 350.)),
    "mantle": lambda vs: (vs * 1.76, 3.4268 + (vs - 4.5) / 4.5, np.full_like(vs, 150.)),
}


def _derive(vs, grp):
    vp, rho, qs = np.zeros_like(vs), np.zeros_like(vs), np.zeros_like(vs)
    for g in set(grp):
        I = np.array([x == g for x in grp])
        vp[I], rho[I], qs[I] = GROUP_RULES[g](vs[I])
    return vp, rho, qs


def perturbed_batch(H, Vs, Vp=None, Rho=None, Qs=None, Grp=None, xtype="Vs", lo=0.999, hi=1.001):
    """model float32 [1+2L', 5, L'] : row 0 unperturbed, rows 1..L' layer i x lo, then x hi
    (layers with h <= 1e-3 dropped as ``_forward`` does, senskernel.py:182-183)."""
    H, Vs = np.asarray(H, float), np.asarray(Vs, float)
    L = H.size
    stacks = []
    for scale, i in [(1.0, -1)] + [(lo, i) for i in range(L)] + [(hi, i) for i in range(L)]:
        vs = Vs.copy()
        vp = None if Vp is None else np.asarray(Vp, float).copy()
        if i >= 0:
            if xtype == "Vs":
                vs[i] *= scale
            elif xtype == "Vp":
                if vp is None:
                    raise ValueError("xtype='Vp' needs an explicit Vp column (senskernel.py:152)")
                vp[i] *= scale
            else:
                raise ValueError(xtype)
        if vp is None or Rho is None or Qs is None:
            dvp, drho, dqs = _derive(vs, Grp)
        vp_ = vp if vp is not None else dvp
        rho_ = np.asarray(Rho, float) if Rho is not None else drho
        qs_ = np.asarray(Qs, float) if Qs is not None else dqs
        keep = H > 1e-3
        stacks.append(np.stack([vp_[keep], vs[keep], rho_[keep], H[keep], 1.0 / qs_[keep]]))
    return np.stack(stacks).astype(np.float32), np.nonzero(H > 1e-3)[0]


def sens_kernel_pert(H, Vs, Vp=None, Rho=None, Qs=None, Grp=None, periods=range(20, 101, 10),
                     wtype="R", xtype="Vs", device=0):
    """dict(phv=[P, L], grv=[P, L], c0=[P], u0=[P]); NaN columns where a perturbed solve failed."""
    kind = {"R": 2, "L": 1}[wtype]
    model, kept = perturbed_batch(H, Vs, Vp, Rho, Qs, Grp, xtype)
    per = np.asarray(list(periods), np.float32)
    c, u, st = _forward.forward_batch(model, per, kind=kind, device=device)
    Lk = kept.size
    Hk = np.asarray(H, float)[kept]
    out = {}
    for name, v in (("phv", c), ("grv", u)):
        vL, vH = v[1:1 + Lk].astype(np.float64), v[1 + Lk:1 + 2 * Lk].astype(np.float64)
        k = (vH - vL) / 0.2 / Hk[:, None]                      # senskernel.py:150
        bad = (st[1:1 + Lk] != 0) | (st[1 + Lk:] != 0)
        k[bad] = np.nan
        full = np.zeros((per.size, np.asarray(H).size))
        full[:, kept] = k.T
        out[name] = full
    out["c0"], out["u0"], out["status"] = c[0], u[0], st
    return out


def sens_kernel_pert_batch(model, periods, wtype="R", nlay=None, lo=0.999, hi=1.001, chunk=256):
    """Finite-difference Vs kernels of MANY stacks on the device (BASELINE configs[4]: "senskernel
    sensitivity evaluation" of every model of a batch).

    model: torch float32 [M, 5, L] rows (vp, vs, rho, h, 1/Qs) on a HIP device - explicit columns,
    so only Vs is scaled (``_perturb`` with explicit Vp/Rho/Qs columns, senskernel.py:160-166);
    periods: torch float32 [P] ascending.  Each stack contributes its unperturbed solve and 2L
    perturbed ones, ``chunk`` stacks (= chunk*(2L+1) solves) per launch.
    Returns dict(phv, grv: float32 [M, P, L] = (v(hi) - v(lo)) / 0.2 / h_i (senskernel.py:150), NaN
    where a perturbed solve failed; c0, u0: [M, P]; status [M])."""
    import torch
    kind = {"R": 2, "L": 1}[wtype]
    M, _, L = model.shape
    P = periods.numel()
    dev = model.device
    # scale[j, i]: factor on layer i of variant j (0: none, 1..L: lo, L+1..2L: hi)
    scale = torch.ones((2 * L + 1, L), dtype=torch.float32, device=dev)
    ar = torch.arange(L, device=dev)
    scale[1 + ar, ar] = lo
    scale[1 + L + ar, ar] = hi
    out = {k: torch.empty((M, P, L), dtype=torch.float32, device=dev) for k in ("phv", "grv")}
    out["c0"] = torch.empty((M, P), dtype=torch.float32, device=dev)
    out["u0"] = torch.empty((M, P), dtype=torch.float32, device=dev)
    out["status"] = torch.empty(M, dtype=torch.int32, device=dev)
    V = 2 * L + 1
    plan = None
    for a in range(0, M, chunk):
        m = model[a:a + chunk]
        n = m.shape[0]
        big = m[:, None].expand(n, V, 5, L).clone()
        big[:, :, 1, :] *= scale[None]
        nl = None if nlay is None else nlay[a:a + chunk].repeat_interleave(V)
        if plan is None or plan.B != n * V:
            plan = _forward.BatchPlan(n * V, L, P, device=dev)
        c, u, st = plan.run(big.reshape(n * V, 5, L), periods, kind=kind, nlay=nl)
        c, u, st = c.view(n, V, P), u.view(n, V, P), st.view(n, V)
        bad = ((st[:, 1:1 + L] != 0) | (st[:, 1 + L:] != 0))[:, None, :]           # [n, 1, L]
        h = m[:, 3, :][:, None, :]
        for name, v in (("phv", c), ("grv", u)):
            k = (v[:, 1 + L:] - v[:, 1:1 + L]).transpose(1, 2) / 0.2 / h         # [n, P, L]
            out[name][a:a + n] = torch.where(bad | (h <= 0), torch.full_like(k, float("nan")), k)
        out["c0"][a:a + n], out["u0"][a:a + n], out["status"][a:a + n] = c[:, 0], u[:, 0], st[:, 0]
    return out


def analytic_kernels(model, periods, wtype="R", nlay=None, want_vp=True, want_rho=True):
    """Sensitivity kernels of a whole batch from ONE forward solve (``surfdisp_forward_kernels_device``):
    the partial derivatives REIGEN / LEIGEN form from their energy integrals and never return
    (surfa.f:1130-1135, 1204-1207; 561-565, 584-585), with the chain factors of the attenuation
    correction and the earth flattening applied so that they refer to the caller's layer values.

    model: torch float32 [M, 5, L] (vp, vs, rho, h, 1/Qs) on a HIP device; periods float32 [P].
    Returns dict(dcdb, dcda, dcdr: float32 [M, P, L] = d c / d (Vs | Vp | rho) per layer (dcda only for
    Rayleigh); c0, u0 [M, P]; status [M]; phv = dcdb * Vs / 100 / h, the reference's ``SensKernelPert``
    units ((v(1.001 Vs) - v(0.999 Vs)) / 0.2 / H, senskernel.py:150))."""
    import torch
    kind = {"R": 2, "L": 1}[wtype]
    M, _, L = model.shape
    plan = _forward.BatchPlan(M, L, periods.numel(), device=model.device)
    c, u, st, kb, ka, kr = plan.run_kernels(model, periods, kind=kind, nlay=nlay, want_vp=want_vp, want_rho=want_rho)
    h = model[:, 3, :][:, None, :]
    vs = model[:, 1, :][:, None, :]
    phv = torch.where(h > 0, kb * vs / 100.0 / torch.where(h > 0, h, torch.ones_like(h)), torch.zeros_like(kb))
    return dict(dcdb=kb, dcda=ka, dcdr=kr, c0=c, u0=u, status=st, phv=phv)


class SensKernelPert:
    """Drop-in for the reference class of the same name (``senskernel.py:129-158``): ``model`` is a CSV
    path or a pandas DataFrame with columns ``H``, ``Vs`` and either ``Vp``/``Rho``/``Qs`` or ``Grp``
    (water / sediment / crust / mantle rules of ``sensModel._convert``, ``:104-124``).  After
    construction ``kernel['Vs']`` (and ``kernel['Vp']`` when the frame has a ``Vp`` column) hold
    float64 [n_periods, n_layers] arrays in the reference's units, (v(1.001 x) - v(0.999 x)) / 0.2 / H;
    ``periods = range(Tmin, Tmax + Tstep//2, Tstep)``.

    ``method='fd'`` reproduces the reference's finite differences (one batched solve of 2L+1 stacks per
    column); ``method='analytic'`` takes the partials of one solve (``surfdisp_forward_kernels_device``)
    and converts them, dc/dx * x / 100 / H - the same quantity without the fp32 differencing noise
    (for a layer whose Vp/Rho/Qs follow from Vs through ``Grp``, the reference's Vs perturbation also moves
    them; the analytic route applies that chain rule).  Love kernels use Love velocities (the reference
    reads ``cr0`` for both wave types and returns ``None`` for Love, SURVEY.md section 4 defect 7)."""

    def __init__(self, model, wtype="R", Tmin=20, Tmax=100, Tstep=10, dz=2, method="fd", device=0):
        import pandas as pd
        if isinstance(model, str):
            df = pd.read_csv(model)
        elif isinstance(model, pd.DataFrame):
            df = model.copy()
        else:
            raise ValueError(f"Wrong model input: {model}")
        self.df = df
        self.wtype = wtype
        self.periods = range(Tmin, Tmax + Tstep // 2, Tstep)
        H = df["H"].to_numpy(float)
        Vs = df["Vs"].to_numpy(float)
        grp = list(df["Grp"]) if "Grp" in df else None
        col = lambda k: df[k].to_numpy(float) if k in df else None
        Vp, Rho, Qs = col("Vp"), col("Rho"), col("Qs")
        self.H, self.Vs = H, Vs
        self.kernel = {}
        if method == "fd":
            self.kernel["Vs"] = sens_kernel_pert(H, Vs, Vp, Rho, Qs, grp, self.periods, wtype, "Vs", device)["phv"]
            if Vp is not None:
                self.kernel["Vp"] = sens_kernel_pert(H, Vs, Vp, Rho, Qs, grp, self.periods, wtype, "Vp", device)["phv"]
        elif method == "analytic":
            import torch
            dVp, dRho, dQs = (None, None, None) if grp is None else _derive(Vs, grp)
            vp = Vp if Vp is not None else dVp
            rho = Rho if Rho is not None else dRho
            qs = Qs if Qs is not None else dQs
            keep = H > 1e-3
            m = np.stack([vp[keep], Vs[keep], rho[keep], H[keep], 1.0 / qs[keep]])[None].astype(np.float32)
            dev = torch.device(f"cuda:{device}")
            per = torch.as_tensor(np.asarray(list(self.periods), np.float32), device=dev)
            out = analytic_kernels(torch.from_numpy(m).to(dev), per, wtype=wtype)
            kb = out["dcdb"][0].double().cpu().numpy()
            ka = out["dcda"][0].double().cpu().numpy() if out["dcda"] is not None else np.zeros_like(kb)
            kr = out["dcdr"][0].double().cpu().numpy()
            Hk, vsk, vpk = H[keep], Vs[keep], vp[keep]
            dvp = np.zeros_like(vsk); drho = np.zeros_like(vsk)      # d(Vp, Rho)/dVs through the Grp rules
            if grp is not None:
                g = np.asarray(grp)[keep]
                if Vp is None:
                    dvp = np.select([g == "sediment", g == "crust", g == "mantle"], [1.23, 1.8, 1.76], 0.0)
                if Rho is None:
                    drho = np.select([g == "sediment", g == "crust", g == "mantle"],
                                     [0.3601 * 1.23, 0.3601 * 1.8, 1.0 / 4.5], 0.0)
            full = np.zeros((len(list(self.periods)), H.size))
            full[:, keep] = (kb + ka * dvp[None, :] + kr * drho[None, :]) * vsk[None, :] / 100.0 / Hk[None, :]
            self.kernel["Vs"] = full
            if Vp is not None:
                fullp = np.zeros_like(full)
                fullp[:, keep] = ka * vpk[None, :] / 100.0 / Hk[None, :]
                self.kernel["Vp"] = fullp
        else:
            raise ValueError("method must be 'fd' or 'analytic'")
