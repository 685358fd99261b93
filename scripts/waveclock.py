#!/usr/bin/env python3
"""GPU box, developer build `make -C pysurfinv_amd/csrc waveclock`: what the wavefronts of the root search do.
    SURFDISP_LIB_PATH=pysurfinv_amd/lib/libsurfdisp_wclk.so python scripts/waveclock.py [team] [B] [L] [model] [kind]
model: synth (SURVEY 8(d) generator, the bench batch at L = 10), mcmc (prior draws of the 96-layer continental model of the
grid leg), c5 (thermal oceanic stacks); kind: 2 Rayleigh / 1 Love.  Per launch: wavefront lifetimes, occupancy over time,
the time split of a wavefront (s_memtime, instrumented: evaluations / end-of-period block / trial choice / team decisions),
team-passes by state, and how many of the lane x layer slots of the evaluations did work (trip count = max over lanes)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import _lib, forward, synth
team = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
Lreq = int(sys.argv[3]) if len(sys.argv) > 3 else 10
which = sys.argv[4] if len(sys.argv) > 4 else "synth"
wave = int(sys.argv[5]) if len(sys.argv) > 5 else 2
L = _lib.lib(); L.surfdisp_set_team(team)
nlay = None
if which == "synth":
    model = torch.from_numpy(synth.synth_models(B, Lreq, seed=0, **({} if Lreq == 10 else {"total_thickness": 220.0}))).cuda()
    per = torch.from_numpy(synth.default_periods(20)).cuda()
else:
    from pysurfinv_amd import settings
    from pysurfinv_amd.brownian import TorchProposer
    from pysurfinv_amd.layers_batch import Model1DBatch
    mb = Model1DBatch(settings.MCMC_SETTING if which == "mcmc" else settings.C5_SETTING, device="cuda:0")
    model, nlay = mb.to_model(TorchProposer(mb.spec, torch.device("cuda:0"), seed=1).reset(B))
    model = model.contiguous()
    per = torch.as_tensor(np.asarray(settings.MCMC_PERIODS, np.float32) if which == "mcmc" else synth.default_periods(20), device="cuda:0")
Lm, P = int(model.shape[2]), int(per.numel())
plan = forward.BatchPlan(B, Lm, P)
buf = torch.zeros(16 * (B + 64) + 1024, dtype=torch.int64, device="cuda")
for kind, name in ((wave, "c+U"), (wave | 0x10, "phase only")):
    plan.run(model, per, kind=kind, nlay=nlay); torch.cuda.synchronize()
    buf.zero_()
    L.surfdisp_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
    plan.run(model, per, kind=kind, nlay=nlay); torch.cuda.synchronize()
    L.surfdisp_debug_buffer(ctypes.c_void_p(0))
    raw = buf[:16 * B].cpu().numpy().reshape(B, 16)
    raw = raw[raw[:, 1] > 0].astype(np.float64)               # the wavefronts that ran
    nw = raw.shape[0]
    G = B * 64 // nw if nw else 0
    t = raw[:, :2] * 1e-2                                                        # 100 MHz ticks -> us
    t0, t1 = t[:, 0].min(), t[:, 1].max()
    life = t[:, 1] - t[:, 0]
    dur = t1 - t0
    ts = np.linspace(t0, t1, 41)
    alive = [(int(((t[:, 0] <= x) & (t[:, 1] > x)).sum())) for x in ts]
    print(f"{which} B={B} L={Lm} P={P} kind={wave} {name}: ~{G} lanes per stack, {nw} wavefronts, kernel span {dur:.0f} us; lifetime mean {life.mean():.0f} "
          f"min {life.min():.0f} q10 {np.quantile(life, .1):.0f} q50 {np.median(life):.0f} q90 {np.quantile(life, .9):.0f} max {life.max():.0f} us; "
          f"start spread {t[:, 0].max() - t0:.0f} us; mean lifetime / span = {life.mean() / dur:.2f}")
    print("   wavefronts alive at 0, 2.5, ... 100 % of the span:", alive)
    cyc, ev, bd, npass, pre = raw[:, 2], raw[:, 3], raw[:, 4], raw[:, 5], raw[:, 6]
    print(f"   per wavefront: {cyc.mean():.3e} cycles in the main loop over {npass.mean():.0f} passes (min {npass.min():.0f} max {npass.max():.0f}), "
          f"{cyc.sum() / npass.sum():.0f} cycles per pass; evaluations {ev.sum() / cyc.sum():.3f}, end-of-period block {bd.sum() / cyc.sum():.3f}, "
          f"trial choice incl. layer dropping {pre.sum() / cyc.sum():.3f}, team decisions {1 - (ev.sum() + bd.sum() + pre.sum()) / cyc.sum():.3f}")
    sc, rf, nv, el, idle = (raw[:, i].sum() for i in (7, 8, 9, 10, 11))
    tp = sc + rf + nv + el + idle
    stacks = B
    print(f"   team-passes per stack: scan {sc / stacks:.1f} refine {rf / stacks:.1f} nevill {nv / stacks:.1f} ellip {el / stacks:.1f} "
          f"idle (team done, wavefront still running) {idle / stacks:.1f}  [shares {sc / tp:.2f} {rf / tp:.2f} {nv / tp:.2f} {el / tp:.2f} {idle / tp:.2f}]")
    trip, ll, lanes = raw[:, 12].sum(), raw[:, 13].sum(), raw[:, 14].sum()
    print(f"   evaluations: {lanes / stacks:.0f} lane-evaluations per stack ({lanes / stacks / P:.1f} per root search), mean layers per evaluation "
          f"{ll / lanes:.1f}, mean trip count of a pass {trip / npass.sum():.1f}; lane x layer slots doing work {ll / (64 * trip):.3f}")
