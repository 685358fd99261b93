#!/usr/bin/env python3
"""GPU box, developer build `make -C pysurfinv_amd/csrc waveclock`: lifetimes of the root-search wavefronts of one
bench batch (B = 65536 x L10 x P20 Rayleigh) - how much of the kernel's duration the machine is full.
    SURFDISP_LIB_PATH=pysurfinv_amd/lib/libsurfdisp_wclk.so python scripts/waveclock.py [team] [B]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import _lib, forward, synth
team = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
L = _lib.lib(); L.surfdisp_set_team(team)
G = L.surfdisp_get_team(B, 10)
model = torch.from_numpy(synth.synth_models(B, 10, seed=0)).cuda(); per = torch.from_numpy(synth.default_periods(20)).cuda()
plan = forward.BatchPlan(B, 10, 20)
nw = B * G // 64
buf = torch.zeros(6 * nw + 1024, dtype=torch.int64, device="cuda")
for kind, name in ((2, "c+U"), (2 | 0x10, "phase only")):
    plan.run(model, per, kind=kind); torch.cuda.synchronize()
    L.surfdisp_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
    plan.run(model, per, kind=kind); torch.cuda.synchronize()
    L.surfdisp_debug_buffer(ctypes.c_void_p(0))
    raw = buf[:6 * nw].cpu().numpy().reshape(nw, 6).astype(np.float64)
    t = raw[:, :2] * 1e-2                                                        # 100 MHz ticks -> us
    t0, t1 = t[:, 0].min(), t[:, 1].max()
    life = t[:, 1] - t[:, 0]
    dur = t1 - t0
    # waves alive over time
    ts = np.linspace(t0, t1, 41)
    alive = [(int(((t[:, 0] <= x) & (t[:, 1] > x)).sum())) for x in ts]
    print(f"{name}: team {G}, {nw} wavefronts, kernel span {dur:.0f} us; lifetime mean {life.mean():.0f} min {life.min():.0f} "
          f"q10 {np.quantile(life, .1):.0f} q50 {np.median(life):.0f} q90 {np.quantile(life, .9):.0f} max {life.max():.0f} us; "
          f"start spread {t[:, 0].max() - t0:.0f} us; mean lifetime / span = {life.mean() / dur:.2f}")
    print("   wavefronts alive at 0, 2.5, ... 100 % of the span:", alive)
    packed = buf[:6 * nw].cpu().numpy().reshape(nw, 6)[:, 5]
    cyc, ev, bd, npass, pre = raw[:, 2], raw[:, 3], raw[:, 4], (packed & 0xFFFFF).astype(np.float64), (packed >> 20).astype(np.float64)
    print(f"   per wavefront (s_memtime, instrumented build): {cyc.mean():.3e} cycles in the main loop over {npass.mean():.0f} passes; "
          f"secular-function evaluations {ev.sum() / cyc.sum():.3f} of them, end-of-period block (store, next period's set-up, stack rebuild) "
          f"{bd.sum() / cyc.sum():.3f}, choice of the trial velocities incl. layer dropping {pre.sum() / cyc.sum():.3f}, team decisions after the "
          f"evaluation {1 - (ev.sum() + bd.sum() + pre.sum()) / cyc.sum():.3f}")
