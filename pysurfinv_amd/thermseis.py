"""Batched thermal -> seismic conversions (reference ``ThermSeis.py``), torch
float64 on the device, one row per chain / grid point.

Reference behaviour followed (``/root/reference/ThermSeis.py``):
* ``TherModel._calP`` (``:22-27``) lithostatic pressure ``3.4e3 * 9.8 * z`` and ``_calRho`` (``:30-35``);
* ``HSCM`` (``:56-101``): half-space cooling geotherm joined to a 0.4 K/km adiabat; the mantle
  temperature ``Tm`` and the junction depth come from a fixed 16-step bisection of a
  finite-difference tangent condition (``:64-79``);
* ``OceanSeisRitz`` (``:103-173``): five-mineral Voigt-Reuss-Hill shear velocity;
* ``OceanSeisYaTa._anel`` (``:325-412``), ``OceanSeisYaTa`` / ``OceanSeisRuan`` (``:414-448``):
  pre-melting anelasticity, Vs and Qs;
* ``OceanSeisBass`` / ``Stix`` / ``PM13`` / ``YaTa_unrelaxed`` and ``behn2009Shear`` (elementwise);
* ``OceanSeisJack.creep10`` is NOT built: it integrates with ``Triforce.mathPlus.logQuad``, a
  dependency that is neither vendored nor pinned (parity unpinned).

All functions take tensors that broadcast against each other: ``age`` ``[B]`` or ``[B, 1]`` and
``zdeps`` ``[N]`` or ``[B, N]`` (km).  Pinned by fixtures captured from the imported reference
(``tests/golden/make_golden_therm.py`` -> ``tests/golden/ref_therm.npz``).
"""
from __future__ import annotations

import math

import torch

C2K = 273.15
YEAR = 365 * 24 * 3600

# elastic parameters of the five minerals, ThermSeis.py:108-129 (Goes et al. / Shapiro & Ritzwoller)
#            rho0      rho_X    K0   K_T     K_P  K_X  mu0  mu_T    mu_P mu_X  alpha0     alpha1      alpha2     alpha3
_MINERALS = (
    (3.222e3, 1.182e3, 129, -16e-3, 4.2, 0,   82,  -14e-3, 1.4, -30, 0.2010e-4, 0.1390e-7,  0.1627e-2, -0.3380),   # olivine
    (3.198e3, 0.804e3, 111, -12e-3, 6.0, -10, 81,  -11e-3, 2.0, -29, 0.3871e-4, 0.0446e-7,  0.0343e-2, -1.7278),   # orthopyroxene
    (3.280e3, 0.377e3, 105, -13e-3, 6.2, 13,  67,  -10e-3, 1.7, -6,  0.3206e-4, 0.0811e-7,  0.1347e-2, -1.8167),   # clinopyroxene
    (3.578e3, 0.702e3, 198, -28e-3, 5.7, 12,  108, -12e-3, 0.8, -24, 0.6969e-4, -0.0108e-7, -3.0799e-2, 5.0395),   # spinel
    (3.565e3, 0.758e3, 173, -21e-3, 4.9, 7,   92,  -10e-3, 1.4, -7,  0.0991e-4, 0.1165e-7,  1.0624e-2, -2.5000),   # garnet
)
_WS_DEFAULT = (0.75, 0.21, 0.035, 0, 0.005)


def _t(x, like=None):
    if torch.is_tensor(x):
        return x.to(torch.float64)
    dev = like.device if torch.is_tensor(like) else None
    if isinstance(x, (int, float)):                          # a fill kernel, not a host->device copy
        return torch.full((), float(x), dtype=torch.float64, device=dev)     # (HIP-graph capturable)
    return torch.as_tensor(x, dtype=torch.float64, device=dev)


class Thermal:
    """zdeps (km), T (K), P (Pa), rho (kg/m^3) - the fields of the reference's ``TherModel``."""
    __slots__ = ("zdeps", "T", "P", "rho", "Tm", "z_adia")

    def __init__(self, zdeps, T, P, rho, Tm=None, z_adia=None):
        self.zdeps, self.T, self.P, self.rho, self.Tm, self.z_adia = zdeps, T, P, rho, Tm, z_adia


def pressure(zdeps, rho=3.4e3):
    """TherModel._calP, ThermSeis.py:22-27."""
    return rho * 9.8 * zdeps * 1000


def density(P, T, rho0=3.42e3, P0=0.6e9, T0=500 + C2K, alpha=4.4e-5, kappa=6.12e-12):
    """TherModel._calRho, ThermSeis.py:30-35."""
    return rho0 * (1 - alpha * (T - T0)) * (1 + kappa * (P - P0))


def hscm_mantle_temperature(age, Tp=1325.0, kappa=1e-6, T0=0.0, Da=0.4):
    """``calTm`` of ``HSCM._calT`` (ThermSeis.py:64-79): bisection on [0, 400] km until the bracket is
    below 0.01 km - always 16 halvings - of ``f/f' - z - (Tp-T0)/Da`` with a forward difference of
    step 0.001 km.  Returns (Tm, z_adiaBegin), shaped like ``age``."""
    age = _t(age)
    Tp = _t(Tp, age)
    scale = 1e3 / (2 * torch.sqrt(age * YEAR * 1 * (kappa / 1e-6)))

    def f(z):
        return torch.erf(z * scale)

    def g(z):
        dz = 0.001
        fz = f(z)
        dfz = (f(z + dz) - fz) / dz + 1e-10
        return fz / dfz - z - (Tp - T0) / Da

    z0 = torch.zeros_like(age * Tp)
    z1 = torch.full_like(z0, 400.0)
    for _ in range(16):                                     # 400 / 2**16 < 0.01 <= 400 / 2**15
        z2 = (z1 + z0) / 2
        neg = g(z2) < 0
        z0 = torch.where(neg, z2, z0)
        z1 = torch.where(neg, z1, z2)
    Tm = (Da * z1 + Tp - T0) / f(z1) + T0
    return Tm, z0


def hscm(age, zdeps=None, rho0=3.43e3, Tp=1325.0, kappa=1e-6):
    """``HSCM(age, zdeps, rho0, Tp, kappa)`` (ThermSeis.py:56-101).  ``age`` [B] (Ma), ``zdeps`` [N]
    or [B, N] km (default ``linspace(0, 200, 200)``).  Returns a ``Thermal`` with [B, N] fields."""
    age = _t(age).reshape(-1, 1)
    if zdeps is None:
        zdeps = torch.linspace(0, 200, 200, dtype=torch.float64, device=age.device)
    zdeps = _t(zdeps, age)
    if zdeps.dim() == 1:
        zdeps = zdeps[None, :]
    zdeps = zdeps.expand(age.shape[0], -1) if zdeps.shape[0] == 1 else zdeps
    Tp = _t(Tp, age)
    Tp = Tp.reshape(-1, 1) if Tp.dim() else Tp
    P = pressure(zdeps)
    T0, Da = 0.0, 0.4
    T_adiabatic = Tp + zdeps * Da
    Tm, z_adia = hscm_mantle_temperature(age, Tp, kappa, T0, Da)
    theta = torch.erf(zdeps * 1e3 / (2 * torch.sqrt(age * YEAR * 1 * (kappa / 1e-6))))
    T = (Tm - T0) * theta + T0
    # from the first grid point deeper than the junction on, the adiabat (ThermSeis.py:90-99; no such
    # point: the conductive profile is kept)
    below = torch.cummax((zdeps > z_adia).to(torch.int8), dim=1).values.bool()
    T = torch.where(below, T_adiabatic.expand_as(T), T) + C2K
    rho = density(P, T, rho0=rho0)
    return Thermal(zdeps, T, P, rho, Tm.reshape(-1), z_adia.reshape(-1))


def ritz_vs(ther: Thermal, X=0.1, ws=_WS_DEFAULT, rho_type="raw"):
    """``OceanSeisRitz(therMod, X=, ws=, RhoType=).vs`` in km/s (ThermSeis.py:132-173); also returns
    the aggregate density and shear modulus (``_rho``, ``_mu``)."""
    T, P = ther.T, ther.P / 1e9
    Tref, Pref = 273.15, 101.325e-6
    ws = [float(w) for w in ws]
    mu_v = K_v = rho_a = 0
    mu_r = K_r = 0
    for w, (rho0, rho_X, K0, K_T, K_P, K_X, mu0, mu_T, mu_P, mu_X, a0, a1, a2, a3) in zip(ws, _MINERALS):
        alpha = a0 + a1 * T + a2 * T ** (-1) + a3 * T ** (-2)
        rho0X = rho0 * rho_X / 1e3 if rho_type == "raw" else rho0 + X * rho_X
        mu = mu0 + (T - Tref) * mu_T + (P - Pref) * mu_P + X * mu_X
        K = K0 + (T - Tref) * K_T + (P - Pref) * K_P + X * K_X
        rho = ther.rho if rho_type == "from_thermal" else rho0X * (1 - alpha * (T - Tref) + (P - Pref) / K)
        rho_a = rho_a + w * rho
        mu_v, mu_r = mu_v + w * mu, mu_r + w / mu
        K_v, K_r = K_v + w * K, K_r + w / K
    mu = 0.5 * (mu_v + 1 / mu_r) * 1e9
    vs = torch.sqrt(mu / rho_a)
    return vs / 1000, rho_a, mu


def solidus(P, Tm="Takei2017"):
    """``calTm`` inside ``OceanSeisYaTa._anel`` (ThermSeis.py:332-346); P in Pa, result in K."""
    Pg = P / 1e9
    if Tm == "Ruan2018":
        return -5.1 * Pg ** 2 + 92.5 * Pg + 1120.6 + C2K
    if Tm == "Hirschmann2009":
        return -5.1 * Pg ** 2 + 132.9 * Pg + 1120.6 + C2K
    if Tm == "Takei2017":
        return 1326 + (Pg * 30 - 50) + C2K
    try:
        return Tm + 1 - 1
    except TypeError:
        raise ValueError(f"Tm = {Tm} is not a numerical variable!")


def anelasticity(T, P, period, Tm="Takei2017"):
    """``OceanSeisYaTa._anel`` (ThermSeis.py:325-412): (J1, J2) of Yamauchi & Takei 2016."""
    A_B, tau_np, alpha = 0.664, 6e-5, 0.38
    Tn = T / solidus(P, Tm)
    one = torch.ones_like(Tn)
    # viscosity reduction A_eta, peak amplitude A_P and width sig_P, ThermSeis.py:359-393
    gamma, Tn_eta = 5, 0.94
    safe = torch.where(Tn > 0, Tn, one)
    a_eta = torch.where(Tn < Tn_eta, one,
                        torch.where(Tn < 1, torch.exp(-(Tn - Tn_eta) / (safe - safe * Tn_eta) * math.log(gamma)),
                                    one / gamma))
    a_p = torch.where(Tn < 0.91, 0.01 * one, torch.where(Tn < 0.96, 0.01 + 0.4 * (Tn - 0.91), 0.03 * one))
    sig_p = torch.where(Tn < 0.92, 4 * one, torch.where(Tn < 1, 4 + 37.5 * (Tn - 0.92), 7 * one))
    # Maxwell time, ThermSeis.py:347-358
    E, R, V, etaR, TR, PR = 4.625e5, 8.314, 7.913e-6, 6.22e21, 1200 + C2K, 1.5e9
    mu_U = (72.45 - 0.01094 * (T - C2K) + 1.75 * P * 1e-9) * 1e9
    eta = etaR * torch.exp(E / R * (1 / T - 1 / TR)) * torch.exp(V / R * (P / T - PR / TR)) * a_eta
    tau_M = eta / mu_U
    tau_ns = period / (2 * math.pi * tau_M)
    lg = torch.log(tau_np / tau_ns) / (math.sqrt(2) * sig_p)
    J1b = A_B * (tau_ns ** alpha) / alpha
    J1p = math.sqrt(2 * math.pi) / 2 * a_p * sig_p * (1 - torch.erf(lg))
    J2b = math.pi / 2 * A_B * (tau_ns ** alpha)
    J2p = math.pi / 2 * (a_p * torch.exp(-(lg ** 2)))
    return 1 + J1b + J1p, J2b + J2p + tau_ns


def yata(ther: Thermal, Tm="Takei2017", period=50, p_coef=1.987):
    """``OceanSeisYaTa(therMod, Tm, period)`` (ThermSeis.py:414-422): (vs, qs, vs_unrelaxed), km/s."""
    T, P = ther.T, ther.P
    Ju = 1 / (72.45 - 0.01094 * (T - C2K) + p_coef * P / 1e9) * 1e-9
    J1, J2 = anelasticity(T, P, period, Tm)
    return 1 / torch.sqrt(ther.rho * Ju * J1) / 1000, J1 / J2, 1 / torch.sqrt(ther.rho * Ju) / 1000


def ruan(ther: Thermal, period=50):
    """``OceanSeisRuan(therMod, period)`` (ThermSeis.py:433-448)."""
    return yata(ther, "Ruan2018", period, p_coef=1.75)


def yata_unrelaxed(ther: Thermal):
    """``OceanSeisYaTa_unrelaxed`` (ThermSeis.py:424-431)."""
    Ju = 1 / (72.45 - 0.01094 * (ther.T - C2K) + 1.987 * ther.P / 1e9) * 1e-9
    return 1 / torch.sqrt(ther.rho * Ju) / 1000


def bass(ther: Thermal):
    """``OceanSeisBass`` (ThermSeis.py:175-181)."""
    Ju = 1 / (66.5 - 0.0136 * (ther.T - C2K - 900) + 1.8 * (ther.P / 1e9 - 0.2)) * 1e-9
    return 1 / torch.sqrt(ther.rho * Ju) / 1000


def stix(ther: Thermal):
    """``OceanSeisStix`` (ThermSeis.py:183-188)."""
    return 4.77 + 0.038 * ther.zdeps / 29.80 - 0.000378 * (ther.T - 300)


def pm13(ther: Thermal, period=1):
    """``OceanSeisPM13`` (ThermSeis.py:283-314), Priestley & McKenzie 2013."""
    T, P = ther.T, ther.P
    Ju = 1 / (72.66 - 0.00871 * T + 2.04 * P / 1e9) * 1e-9
    E, Va, R, Pr, Tr = 402.9e3, 7.81e-6, 8.314, 1.5e9, 1473
    eta0 = 10 ** 22.38
    a_star = torch.exp((E + Pr * Va) / (R * Tr) - (E + P * Va) / (R * T))
    f_prime = Ju * (eta0 / a_star) * 1 / period
    x = torch.log(f_prime)
    F = torch.zeros_like(x)
    for c in (3.9461e-9, -3.4761e-7, 9.9473e-6, -5.7175e-5, -2.3616e-3, 0.054332, 0.55097):
        F = F * x + c
    F = torch.where(f_prime > 1e13, torch.ones_like(F), F)
    return 1 / torch.sqrt(ther.rho * (Ju / F)) / 1000


def behn2009_shear(freq, d, T, P, coh=100):
    """``behn2009Shear`` (ThermSeis.py:451-482): (Qinv, shearFactor); T in deg C, P in GPa."""
    T = _t(T) + 273.1
    P = _t(P, T)
    pqref, pq, Tqref, dqref = 1.09, 1, 1265, 1.24e-5
    Eqref, Vqref, Bo, Eq, Vq = 505e3, 1.2e-5, 1.28e8, 420e3, 1.2e-5
    cohref, R, Pqref, rq, alpha = 50, 8.314, 300e6, 1.2, 0.27
    B = Bo * dqref ** (pq - pqref) * (coh / cohref) ** rq * math.exp(((Eq + Pqref * Vq) - (Eqref + Pqref * Vqref)) / R / Tqref)
    Qinv = (B * d ** (-1 * pq) / freq * torch.exp(-(Eq + P * 1e9 * Vq) / R / T)) ** alpha
    F = (1 / math.tan(math.pi * alpha / 2)) / 2
    return Qinv, (1 - F * Qinv) ** 2


def melt_start(age, z_crust):
    """``meltStart(age) - crustH`` of ``OceanMantleHybrid._calVs`` (layers.py:312-320, 340): first
    depth of the DEFAULT half-space model (200 points to 200 km, Tp = 1325) hotter than 0.92 x the
    damp solidus; the last depth when there is none."""
    ther = hscm(age)
    hot = ther.T > 0.92 * solidus(ther.P, "Ruan2018")
    first = torch.argmax(hot.to(torch.int8), dim=1)
    first = torch.where(hot.any(dim=1), first, torch.full_like(first, hot.shape[1] - 1))
    return torch.gather(ther.zdeps, 1, first[:, None]).squeeze(1) - z_crust


def tridiagonal_solve(lo, di, up, rhs):
    """Row-wise solve of tridiagonal systems [B, N] (lo[:, 0] and up[:, -1] ignored) by parallel cyclic
    reduction: ceil(log2 N) sweeps of elementwise tensor ops - a handful of launches, no pivot search,
    no host synchronisation (HIP-graph capturable).  The spline systems it is used for are diagonally
    dominant except for the two not-a-knot end rows, which one reduction step makes dominant."""
    B, N = di.shape
    lo = lo.clone(); up = up.clone()
    lo[:, 0] = 0
    up[:, -1] = 0
    F = torch.nn.functional
    s = 1
    while s < N:
        # neighbours at distance s; beyond the ends: identity rows (lo = up = rhs = 0, di = 1)
        sh = lambda a, fill: (F.pad(a, (s, 0), value=fill)[:, :N], F.pad(a, (0, s), value=fill)[:, s:])
        lo_m, lo_p = sh(lo, 0.0)
        up_m, up_p = sh(up, 0.0)
        di_m, di_p = sh(di, 1.0)
        r_m, r_p = sh(rhs, 0.0)
        al = -lo / di_m
        ga = -up / di_p
        di = di + al * up_m + ga * lo_p
        rhs = rhs + al * r_m + ga * r_p
        lo = al * lo_m
        up = ga * up_p
        s *= 2
    return rhs / di


def cubic_spline_through(x, y, keep):
    """Row-wise ``scipy.interpolate.CubicSpline(x[keep], y[keep])(x)`` (not-a-knot ends; extrapolating
    with the end pieces) - ``merge2`` of ``OceanMantleHybrid._calVs`` (layers.py:321-325).
    x, y [B, N] (x ascending), keep bool [B, N] with at least 2 knots per row.  No host
    synchronisation anywhere (HIP-graph capturable)."""
    B, N = x.shape
    dev = x.device
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
    n = keep.sum(dim=1)                                      # knots per row
    n = n.clamp(min=2)                                       # fewer than two knots cannot occur in merge2
    xk = torch.gather(x, 1, order)
    yk = torch.gather(y, 1, order)
    idx = torch.arange(N, device=dev)[None, :]
    # pad behind the last knot by repeating it at unit spacing so that the padded rows stay regular
    last = (n - 1)[:, None]
    xl = torch.gather(xk, 1, last)
    yl = torch.gather(yk, 1, last)
    pad = idx > last
    xk = torch.where(pad, xl + (idx - last).to(x.dtype), xk)
    yk = torch.where(pad, yl, yk)
    dx = xk[:, 1:] - xk[:, :-1]                              # [B, N-1]
    slope = (yk[:, 1:] - yk[:, :-1]) / dx
    # equations for the knot derivatives s (de Boor, "A practical guide to splines", ch. IV):
    #   dx[i] s[i-1] + 2 (dx[i-1] + dx[i]) s[i] + dx[i-1] s[i+1] = 3 (dx[i] slope[i-1] + dx[i-1] slope[i])
    lo = torch.zeros((B, N), dtype=x.dtype, device=dev)      # A[i, i-1]
    di = torch.ones((B, N), dtype=x.dtype, device=dev)       # A[i, i]
    up = torch.zeros((B, N), dtype=x.dtype, device=dev)      # A[i, i+1]
    rhs = torch.zeros((B, N), dtype=x.dtype, device=dev)
    lo[:, 1:-1] = dx[:, 1:]
    di[:, 1:-1] = 2 * (dx[:, :-1] + dx[:, 1:])
    up[:, 1:-1] = dx[:, :-1]
    rhs[:, 1:-1] = 3 * (dx[:, 1:] * slope[:, :-1] + dx[:, :-1] * slope[:, 1:])
    # first row: not-a-knot at knot 1
    d = xk[:, 2] - xk[:, 0]
    di[:, 0] = dx[:, 1]
    up[:, 0] = d
    rhs[:, 0] = ((dx[:, 0] + 2 * d) * dx[:, 1] * slope[:, 0] + dx[:, 0] ** 2 * slope[:, 1]) / d
    # row of the last knot: not-a-knot at knot n-2
    g1 = lambda a, j: torch.gather(a, 1, j.clamp(min=0)[:, None]).squeeze(1)
    li = n - 1
    dxa, dxb = g1(dx, li - 1), g1(dx, li - 2)                # dx[n-2], dx[n-3]
    d = g1(xk, li) - g1(xk, li - 2)
    sa, sb = g1(slope, li - 1), g1(slope, li - 2)
    at_last = idx == last
    lo = torch.where(at_last, d[:, None], lo)
    di = torch.where(at_last, dxb[:, None], di)
    up = torch.where(at_last, torch.zeros_like(up), up)
    rhs = torch.where(at_last, ((dxa ** 2 * sb + (2 * d + dxa) * dxb * sa) / d)[:, None], rhs)
    # padding rows behind the last knot: identity (their values are never used)
    lo = torch.where(pad, torch.zeros_like(lo), lo)
    di = torch.where(pad, torch.ones_like(di), di)
    up = torch.where(pad, torch.zeros_like(up), up)
    rhs = torch.where(pad, torch.zeros_like(rhs), rhs)
    # rows with 3 knots: the parabola through them (both not-a-knot conditions coincide); 2: a line.
    # Blended in with masks (no host synchronisation: the whole path is HIP-graph capturable).
    one, zero = torch.ones_like(d), torch.zeros_like(d)
    three = (n == 3)[:, None]
    st = lambda *cols: torch.stack(cols, dim=1)
    lo[:, :3] = torch.where(three, st(zero, dx[:, 1], one), lo[:, :3])
    di[:, :3] = torch.where(three, st(one, 2 * (dx[:, 0] + dx[:, 1]), one), di[:, :3])
    up[:, :3] = torch.where(three, st(one, dx[:, 0], zero), up[:, :3])
    rhs[:, :3] = torch.where(three, st(2 * slope[:, 0], 3 * (dx[:, 0] * slope[:, 1] + dx[:, 1] * slope[:, 0]),
                                       2 * slope[:, 1]), rhs[:, :3])
    two = (n == 2)[:, None]
    lo[:, :2] = torch.where(two, st(zero, zero), lo[:, :2])
    di[:, :2] = torch.where(two, st(one, one), di[:, :2])
    up[:, :2] = torch.where(two, st(zero, zero), up[:, :2])
    rhs[:, :2] = torch.where(two, st(slope[:, 0], slope[:, 0]), rhs[:, :2])
    s = tridiagonal_solve(lo, di, up, rhs)
    # piece i on [xk[i], xk[i+1]]:  y = y_i + s_i t + c2 t^2 + c3 t^3
    tq = (s[:, :-1] + s[:, 1:] - 2 * slope) / dx
    c3 = tq / dx
    c2 = (slope - s[:, :-1]) / dx - tq
    # interval of every evaluation point (clamped: extrapolation uses the end pieces)
    piece = torch.searchsorted(xk.contiguous(), x.contiguous(), right=True) - 1
    piece = torch.minimum(piece.clamp(min=0), (n - 2)[:, None])
    g = lambda a: torch.gather(a, 1, piece)
    t = x - g(xk)
    return g(yk) + t * (g(s) + t * (g(c2) + t * g(c3)))
