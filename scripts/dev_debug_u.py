#!/usr/bin/env python3
import os, sys, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hostcheck"))
from conftest import load_cases
from pysurfinv_amd import _lib, forward
import torch
from run_hostcheck import oracle_dbg, host_group
np.set_printoptions(linewidth=250, precision=7)
cases = load_cases()
for name in sys.argv[1:] or ["two_layer_R", "synth_L5_R"]:
    d = cases[name]
    B, _, L = d["model"].shape; P = len(d["periods"])
    per = np.ascontiguousarray(d["periods"], np.float32)
    plan = forward.BatchPlan(B, L, P)
    dbg = torch.zeros(B * P * 16, dtype=torch.float64, device='cuda')
    _lib.lib().surfdisp_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
    c, u, st = plan.run(torch.from_numpy(d["model"]).cuda(), torch.from_numpy(per).cuda(), kind=d["kind"])
    torch.cuda.synchronize()
    c = c.cpu().numpy(); u = u.cpu().numpy()
    off = ((10 * L * B * 4 + 255) // 256) * 256
    ratio = plan.workspace[off: off + P * B * 4].view(torch.float32).cpu().numpy().reshape(P, B).T.copy()
    co, uo, ro = oracle_dbg(d["model"], per, d["kind"])
    e = np.abs(u / uo - 1)
    b = np.unravel_index(e.argmax(), e.shape)[0]
    print(name, "worst stack", b)
    print(" per   ", per)
    print(" errU  ", e[b])
    print(" errC  ", np.abs(c[b] / co[b] - 1))
    print(" errR  ", np.abs(ratio[b] / ro[b] - 1))
    print(" ratio ", ratio[b]); print(" ratioO", ro[b])
    # host-compiled group math fed with the GPU's c and ratio
    hd = np.zeros((B, P, 16)); uh = host_group(d["model"], per, d["kind"], c, ratio, hd)
    gd = dbg.cpu().numpy().reshape(B, P, 16)
    k0 = int(np.nanargmax(e[b]))
    print(' k0', k0, '\n GPU dbg', gd[b, k0], '\n HOSTdbg', hd[b, k0])
    print(" errU(host math on GPU c,ratio vs oracle)", np.abs(uh[b] / uo[b] - 1))
    print(" errU(GPU vs host math on same inputs)   ", np.abs(u[b] / uh[b] - 1))
