#!/usr/bin/env python3
"""GPU-only soak: the opt-in fast scan (SURFDISP_FASTSCAN) against the default point-by-point scan, bit for bit, on random stacks.
SOAK_MONO=1: monotone stacks only (Vs, Vp non-decreasing with depth)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import _lib, forward, synth

L = _lib.lib()
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "0")))
T_END = time.time() + float(os.environ.get("SOAK_SECONDS", "120"))
MONO = os.environ.get("SOAK_MONO", "1") == "1"
ONLY = os.environ.get("SOAK_ONLY")                      # 'wild' / 'sediment': that family only (sediment: wider ranges)
VSMIN = float(os.environ.get("SOAK_WILD_VSMIN", "0.1"))
T_LAST = time.time()
nst = nval = ndif = npat = ncase = nbig = 0
byfam = {}
# realistic stacks: uniform prior draws of the continental and the thermal oceanic parametrisation (tests/golden)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from settings import CONT
from settings_therm import HYBRID_STATIC
from pysurfinv_amd.layers_batch import Model1DBatch
from pysurfinv_amd.mcmc import MetropolisBatch
PRIOR = {}
for nm, st in (("prior_cont", CONT), ("prior_ocean", HYBRID_STATIC)):
    mb = Model1DBatch(st, device="cuda:0")
    PRIOR[nm] = (mb, MetropolisBatch(mb.spec, mb.to_model, np.array([10., 20.]), np.array([3.5, 3.6]), np.array([.01, .01]),
                                     device="cuda:0", seed=int(os.environ.get("SOAK_SEED", "0"))))
worst = []
saved = []
while time.time() < T_END:
    plo, phi = 3.0, 150.0; fam = 'synth'
    Ln = int(rng.integers(2, 48)); B = 4096; kind = int(rng.integers(1, 3))
    noise = float(rng.choice([0.02, 0.05, 0.1, 0.2])); tt = float(rng.choice([30., 60., 120., 200., 400.]))
    mono = True if MONO else bool(rng.random() < 0.5)
    m = synth.synth_models(B, Ln, seed=int(rng.integers(1 << 30)), noise=noise, monotone=mono, total_thickness=tt)
    if ONLY is None and rng.random() < 0.2 and Ln >= 4:
        m[:, 1, 0] = 0.0; m[:, 0, 0] = 1.475; m[:, 2, 0] = 1.027; m[:, 4, 0] = 1e-4; m[:, 3, 0] = rng.uniform(0.3, 4.0, B); fam = 'water'
    elif (ONLY is None and rng.random() < 0.3 and Ln >= 4) or (ONLY == 'sediment' and Ln >= 4):
        # soft sediments over rock: strong contrast, fundamental and first overtone nearly touch (osculation)
        m = synth.sediment_models(B, Ln, seed=int(rng.integers(1 << 30)), noise=noise, total_thickness=tt,
                                  max_layers=int(rng.choice([4, 4, 12])), water=bool(rng.random() < 0.3))
        plo, phi = 0.3, 30.0; fam = 'sediment'
        if ONLY == 'sediment':                              # thicker and stiffer sediments, longer periods
            f = rng.uniform(1.0, 3.0, (B, 1)); g = rng.uniform(1.0, 1.8, (B, 1))
            rock = m[:, 1, :] > 2.9
            m[:, 3, :] = np.where(rock, m[:, 3, :], m[:, 3, :] * f)
            m[:, 1, :] = np.where(rock | (m[:, 1, :] == 0.0), m[:, 1, :], np.minimum(m[:, 1, :] * g, 2.85))
            m[:, 1, :] = np.sort(m[:, 1, :], axis=1); m[:, 0, :] = np.sort(np.maximum(m[:, 0, :], 1.5 * m[:, 1, :]), axis=1)
            plo, phi = 0.5, 60.0
    elif ONLY == 'wild' or rng.random() < 0.3:
        # anything monotone: velocities 0.1-5 km/s, thicknesses 10 m - 50 km and periods 0.1-300 s log-uniform,
        # Vp/Vs 1.5-8, sometimes under water
        vs = np.sort(np.exp(rng.uniform(np.log(VSMIN), np.log(5.0), (B, Ln))), axis=1)
        vp = np.sort(vs * np.exp(rng.uniform(np.log(1.5), np.log(8.0), (B, Ln))), axis=1)
        m[:, 1, :] = vs; m[:, 0, :] = vp; m[:, 2, :] = np.sort(rng.uniform(1.6, 3.4, (B, Ln)), axis=1)
        m[:, 3, :] = np.exp(rng.uniform(np.log(0.01), np.log(50.0), (B, Ln))); m[:, 3, -1] = 0.0
        m[:, 4, :] = rng.choice([1e-4, 1 / 600., 1 / 80., 1 / 20.], (B, Ln))
        if rng.random() < 0.3 and Ln >= 3:
            m[:, 1, 0] = 0.0; m[:, 0, 0] = 1.475; m[:, 2, 0] = 1.027; m[:, 4, 0] = 1e-4; m[:, 3, 0] = rng.uniform(0.05, 5.0, B)
            m[:, 0, 1:] = np.maximum(m[:, 0, 1:], 1.475); m[:, 2, 1:] = np.maximum(m[:, 2, 1:], 1.1)
        plo, phi = 0.1, 300.0; fam = 'wild'
    nl_t = None
    if ONLY is None and rng.random() < 0.15:
        fam = str(rng.choice(list(PRIOR))); mb, mc = PRIOR[fam]
        md_t, nl_t = mb.to_model(mc.reset(B)); md_t = md_t.contiguous()
        m = md_t.cpu().numpy(); Ln = m.shape[2]
        plo, phi = (5.0, 100.0) if fam == "prior_cont" else (4.0, 40.0)
    P = int(rng.integers(1, 40))
    per = np.sort(np.exp(rng.uniform(np.log(plo), np.log(phi), P)) if fam == 'wild' else rng.uniform(plo, phi, P)).astype(np.float32)
    team = int(rng.choice([2, 4, 8])); L.surfdisp_set_team(team)
    md = torch.from_numpy(m).cuda(); pd = torch.from_numpy(per).cuda()
    kw = {} if nl_t is None else {'nlay': nl_t}
    indep = bool(rng.random() < 0.15)
    if indep: kw['independent'] = True; fam = fam + '/indep'
    plan = forward.BatchPlan(B, Ln, P)
    c0, u0, s0 = plan.run(md, pd, kind=kind | 0x10, **kw); c0 = c0.clone(); s0 = s0.clone()       # default: every grid point
    c1, u1, s1 = plan.run(md, pd, kind=kind | 0x10, fast_scan=True, **kw)                          # opt-in SURFDISP_FASTSCAN
    d = (c0 != c1)
    nd = int(d.sum()); ndif += nd; nval += c0.numel(); nst += B; ncase += 1
    # another ROOT (not the last bits: a stack the two scans hand to different kernels - production / exact fallback -
    # differs at the 1e-7 level)
    nbig += int(((c0 - c1).abs() > 1e-5 * c0.abs().clamp(min=1e-3)).sum())
    npat += int(((c0 > 0) != (c1 > 0)).any(dim=1).sum())
    f = byfam.setdefault((fam, kind), [0, 0, 0]); f[0] += c0.numel(); f[1] += nd; f[2] += int(((c0 > 0) != (c1 > 0)).sum())
    if nd:
        worst.append((nd, Ln, kind, noise, tt, P, team, float((c0 - c1).abs().max())))
        if len(saved) < 40:
            rows = torch.nonzero(d.any(dim=1)).flatten()[:4].cpu().numpy()
            for r in rows:
                saved.append(dict(model=m[r], per=per, kind=kind, team=team, c_default=c0[r].cpu().numpy(), c_fast=c1[r].cpu().numpy()))
    if time.time() - T_LAST > 45:
        T_LAST = time.time()
        print(f"  ... {ncase} cases, {nst} stacks, {nval} phase velocities, {ndif} differ, {npat} stacks with another zero pattern", flush=True)
L.surfdisp_set_team(0)
if saved:
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    np.save(os.path.join(ROOT, 'gpurun_out', 'scanfail.npy'), np.array(saved, dtype=object), allow_pickle=True)
print(f"scan soak, fast vs default scan (monotone only: {MONO}): {ncase} cases, {nst} stacks, {nval} phase velocities; {ndif} differ ({ndif / max(nval, 1):.2e}); "
      f"{nbig} of them by more than 1e-5 relative (another root); {npat} stacks with a different zero pattern")
for k, v in sorted(byfam.items()):
    print(f"   family {k[0]:9s} kind {k[1]}: {v[0]} values, {v[1]} differ, {v[2]} of them zero/non-zero")
for w in sorted(worst, reverse=True)[:15]:
    print("   differ %d: L=%d kind=%d noise=%.2f thick=%.0f P=%d team=%d max|dc|=%.2e" % w)
