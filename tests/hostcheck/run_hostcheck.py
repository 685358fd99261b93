#!/usr/bin/env python3
"""Debug aid (CPU container): group-velocity device math compiled for the host vs the oracle."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_cases
from oracle import cport

H = ctypes.CDLL(os.environ.get("SURFDISP_HOSTCHECK_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhostcheck.so"))
O = cport.lib()
fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def oracle_dbg(model, per, kind):
    B, _, L = model.shape; P = len(per)
    c = np.zeros((B, P), np.float32); u = np.zeros((B, P), np.float32); r = np.zeros((B, P), np.float32)
    for i in range(B):
        m = np.ascontiguousarray(model[i])
        O.surfdisp_oracle_forward_dbg(L, kind, fp(m[0]), fp(m[1]), fp(m[2]), fp(m[3]), fp(m[4]),
                                      fp(per), P, fp(c[i]), fp(u[i]), fp(r[i]))
    return c, u, r


def host_group(model, per, kind, c, ratio, dbg=None):
    B, _, L = model.shape; P = len(per)
    u = np.zeros((B, P), np.float32)
    model = np.ascontiguousarray(model, np.float32)
    H.sd_hostcheck_group(B, L, None, fp(model), P, fp(per), kind, fp(np.ascontiguousarray(c)),
                         fp(np.ascontiguousarray(ratio)), fp(u),
                         dbg.ctypes.data_as(ctypes.c_void_p) if dbg is not None else None)
    return u


if __name__ == "__main__":
    cases = load_cases()
    sel = sys.argv[1:] or sorted(cases)
    for name in sel:
        d = cases[name]
        per = np.ascontiguousarray(d["periods"], np.float32)
        c, u, r = oracle_dbg(d["model"], per, d["kind"])
        uh = host_group(d["model"], per, d["kind"], c, r)
        ok = u != 0
        err = np.abs(uh[ok] / u[ok] - 1) if ok.any() else np.zeros(1)
        print(f"{name:24s} max {err.max():9.2e}  mean {err.mean():9.2e}")
        if err.max() > 1e-4 and len(sel) < 4:
            e = np.zeros_like(u); e[ok] = np.abs(uh[ok] / u[ok] - 1)
            b = np.unravel_index(e.argmax(), e.shape)[0]
            print(" worst stack", b, "\n per", per, "\n err", e[b], "\n u", u[b], "\n uh", uh[b], "\n c", c[b])
