"""Synthetic layered-earth stacks used by bench.py and the parity tests.

Generator of SURVEY.md section 8(d): MCMC-like perturbed monotone stacks whose
property rules follow the reference's layer classes (Vp = 1.76 Vs, layers.py:261;
rho = 0.541 + 0.3601 Vp, layers.py:234; Qs = 600 / 150, layers.py:187,263).
Pure numpy; no reference code involved.
"""
from __future__ import annotations

import numpy as np

FIELDS = ("vp", "vs", "rho", "h", "qsinv")  # row order of a model[B,5,L] array
                                            # = Fortran argument order a,b,rho,d,qs (fast_surf.f:2-5)


def default_periods(P: int = 20) -> np.ndarray:
    return np.linspace(8.0, 100.0, P).astype(np.float32)


def synth_models(B: int, L: int, seed: int = 0, noise: float = 0.03, monotone: bool = True,
                 total_thickness: float = 200.0) -> np.ndarray:
    """float32 [B, 5, L] batch, rows (vp, vs, rho, h, qsinv); last layer = half-space."""
    rng = np.random.default_rng(seed)
    z = np.linspace(0.0, 1.0, L)
    vs = 3.0 + 1.6 * z + rng.normal(0.0, noise, (B, L))
    if monotone:
        vs = np.sort(vs, axis=1)
    h = np.full((B, L), total_thickness / L)
    vp = 1.76 * vs
    rho = 0.541 + 0.3601 * vp
    qs = np.where(vs < 4.0, 600.0, 150.0)
    return np.stack([vp, vs, rho, h, 1.0 / qs], axis=1).astype(np.float32)


def sediment_models(B: int, L: int, seed: int = 0, noise: float = 0.05, total_thickness: float = 120.0,
                    max_layers: int = 4, water: bool = False) -> np.ndarray:
    """Soft sediments (Vs 0.2-1.4 km/s, Vp/Vs 1.8-3.5, one to four layers of 0.2-3 km) over the monotone rock
    stack of ``synth_models``: fundamental and first higher Rayleigh mode come within 0.02 km/s of each other
    and the secular function carries e^{kd} factors of many orders of magnitude -- the hard case for the
    opt-in fast scan (tests/test_gpu_parity.py, scripts/soak_scan.py)."""
    rng = np.random.default_rng(seed)
    m = synth_models(B, L, seed=seed + 1, noise=noise, monotone=True, total_thickness=total_thickness)
    ns = int(rng.integers(1, min(max_layers, L - 1) + 1))
    vs = np.sort(rng.uniform(0.2, 1.4 if max_layers <= 4 else 3.0, (B, ns)), axis=1)
    m[:, 1, :ns] = vs
    m[:, 0, :ns] = np.sort(np.minimum(vs * rng.uniform(1.8, 3.5, (B, 1)), m[:, 0, ns:ns + 1]), axis=1)
    m[:, 2, :ns] = rng.uniform(1.8, 2.3, (B, ns))
    m[:, 3, :ns] = rng.uniform(0.2, 3.0, (B, ns)) * min(1.0, 4.0 / ns)
    if water and L >= 3:                                   # ``max_layers`` > 4: a gradient of thinner layers up to 3 km/s
        m[:, 1, 0] = 0.0; m[:, 0, 0] = 1.475; m[:, 2, 0] = 1.027; m[:, 4, 0] = 1e-4
        m[:, 3, 0] = rng.uniform(0.05, 5.0, B)
        m[:, 0, 1:] = np.maximum(m[:, 0, 1:], 1.48)
    return m


def water_models(B: int, seed: int = 5) -> np.ndarray:
    """Ocean stacks with a water top layer, 9 layers."""
    rng = np.random.default_rng(seed)
    h = np.array([2, .5, 3.5, 3.5, 20, 30, 50, 80, 0.])
    vs = np.array([0, 1.0, 3.3, 3.9, 4.4, 4.35, 4.4, 4.5, 4.6])
    vp = 1.76 * vs
    vp[0] = 1.475
    vp[1] = 1.23 * vs[1] + 1.28
    rho = 0.541 + 0.3601 * vp
    rho[0] = 1.027
    qs = np.array([1e4, 80, 350, 350, 150, 150, 150, 150, 150.])
    m = np.stack([vp, vs, rho, h, 1.0 / qs])[None].astype(np.float32)
    mm = np.repeat(m, B, 0)
    mm[:, 1, 1:] += rng.normal(0, 0.05, (B, 8)).astype(np.float32)
    mm[:, 3, 0] = rng.uniform(0.5, 4.5, B)
    return mm
