#!/usr/bin/env python3
"""GPU box: the coarse scans - Love's certified one (default since r03) and, with SOAK_KIND=2 and SURFDISP_FASTSCAN=1 in the environment,
Rayleigh's opt-in count-guided one (r04) - against the point-by-point scan
(SURFDISP_EXACTSCAN) on random stacks - the two must agree BIT FOR BIT (same brackets, same refinement); any differing stack is a
failed certificate.  SOAK_SECONDS, SOAK_SEED, SOAK_KIND (1 Love, 2 Rayleigh), SOAK_CU=1: c+U calls instead of phase-only ones, SOAK_DEEP=1: stacks of up to 96 layers (default: below 48)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pysurfinv_amd import _lib, forward, synth
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "0")))
T_END = time.time() + float(os.environ.get("SOAK_SECONDS", "120"))
KIND = int(os.environ.get('SOAK_KIND', '1'))
FLAGS = 0 if os.environ.get('SOAK_CU') == '1' else _lib.PHASE_ONLY
nstack = ncase = nbad = 0
dump = []
t_last = time.time()
while time.time() < T_END:
    L = int(rng.integers(2, 97 if os.environ.get('SOAK_DEEP') == '1' else 48)); B = int(rng.integers(64, 2048)) * (16 if L < 48 else 4)
    noise = float(rng.choice([0.02, 0.05, 0.1, 0.2])); mono = bool(rng.random() < 0.5)
    fam = rng.random()
    if fam < 0.25 and L >= 4:
        model = synth.sediment_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise, total_thickness=float(rng.choice([30., 60., 120., 200., 400.])))
    else:
        model = synth.synth_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise, monotone=mono,
                                   total_thickness=float(rng.choice([20., 60., 120., 200., 400., 800.])))
    if rng.random() < 0.2 and L >= 4:                       # water on top
        model[:, 1, 0] = 0.0; model[:, 0, 0] = 1.475; model[:, 2, 0] = 1.027; model[:, 4, 0] = 1e-4
    P = int(rng.integers(1, 40))
    per = np.sort(rng.uniform(2.0, 150.0, P)).astype(np.float32)
    nlay = None
    if rng.random() < 0.3:
        nlay = rng.integers(2, L + 1, B).astype(np.int32)
    team = int(rng.choice([0, 0, 1, 2, 4, 8]))
    _lib.lib().surfdisp_set_team(team)
    dm, dp = torch.from_numpy(model).cuda(), torch.from_numpy(per).cuda()
    dn = None if nlay is None else torch.from_numpy(nlay).cuda()
    plan = forward.BatchPlan(B, L, P)
    indep = bool(rng.random() < 0.15)
    c1, u1, s1 = (t.clone() for t in plan.run(dm, dp, kind=KIND | FLAGS, nlay=dn, independent=indep))
    c0, u0, s0 = plan.run(dm, dp, kind=KIND | FLAGS | _lib.EXACTSCAN, nlay=dn, independent=indep)
    torch.cuda.synchronize()
    diff = ((c1 != c0).any(dim=1) | (s1 != s0))
    if not (FLAGS & _lib.PHASE_ONLY):                        # c+U: the group velocities too (NaN = NaN)
        diff = diff | ((u1 != u0) & ~(torch.isnan(u1) & torch.isnan(u0))).any(dim=1)
    nb = int(diff.sum())
    nstack += B; ncase += 1; nbad += nb
    if nb:
        i = int(diff.nonzero()[0, 0])
        if os.environ.get("SOAK_DUMP") and len(dump) < 400:
            for q in diff.nonzero()[:, 0].cpu().numpy()[:16]:
                m48 = np.zeros((5, 48), np.float32); m48[:, :L] = model[q]
                p40 = np.zeros(40, np.float32); p40[:P] = per
                pad = lambda a: np.pad(a, (0, 40 - P))
                dump.append(dict(model=m48, nlay=int(nlay[q]) if nlay is not None else L, per=p40, P=P, team=team, indep=int(indep),
                                 c1=pad(c1[q].cpu().numpy()), c0=pad(c0[q].cpu().numpy())))
        print(f"  DIFFERS: L={L} B={B} P={P} team={team} indep={indep} noise={noise} mono={mono}: {nb} stacks; first {i}: c {c1[i].cpu().numpy()} vs {c0[i].cpu().numpy()}", flush=True)
    if time.time() - t_last > 45:
        print(f"  ... {ncase} cases, {nstack} stacks, {nbad} differing", flush=True); t_last = time.time()
_lib.lib().surfdisp_set_team(0)
if dump:
    np.savez_compressed(os.environ['SOAK_DUMP'], **{k: np.array([d[k] for d in dump]) for k in dump[0]})
print(f"{'certified Love' if KIND == 1 else 'count-guided Rayleigh (SURFDISP_FASTSCAN)'} scan vs point-by-point scan: {ncase} cases, {nstack} stacks, {nbad} differing stacks")
sys.exit(1 if nbad else 0)
