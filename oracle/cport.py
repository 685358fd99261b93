"""ctypes wrapper for oracle/libsurfdisp_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never by the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SURFDISP_ORACLE_LIB: another build of the same source (the sanitizer build of `make -C oracle asan`, scripts/sanitize_cpu.sh)
_SO = os.environ.get("SURFDISP_ORACLE_LIB") or os.path.join(_HERE, "libsurfdisp_oracle.so")
_lib = None

OK, PARTIAL, NOROOT, NEVILL, EINVAL = 0, 1, 2, 3, -1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "surfdisp_oracle.c")
    if os.environ.get("SURFDISP_ORACLE_LIB"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libsurfdisp_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        L.surfdisp_oracle_forward.restype = ctypes.c_int
        L.surfdisp_oracle_forward.argtypes = [ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp, fp,
                                              ctypes.c_int, fp, fp, ip,
                                              ctypes.POINTER(ctypes.c_long)]
        L.surfdisp_oracle_forward_batch.restype = ctypes.c_int
        L.surfdisp_oracle_forward_batch.argtypes = [ctypes.c_int, ctypes.c_int, ip, fp, ctypes.c_int,
                                                    fp, ctypes.c_int, fp, fp, ip, ctypes.c_int]
        L.surfdisp_oracle_forward_at.restype = ctypes.c_int
        L.surfdisp_oracle_forward_at.argtypes = [ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp, fp,
                                                 ctypes.c_int, fp, fp, fp]
        dp = ctypes.POINTER(ctypes.c_double)
        L.surfdisp_oracle_partials.restype = ctypes.c_int
        L.surfdisp_oracle_partials.argtypes = [ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp, ctypes.c_float,
                                               dp, dp, dp, dp, ip, ip, fp, fp]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def forward(vp, vs, rho, h, qsinv, periods, kind):
    """One solve -> dict(c, u, status, nsolved, n_delta)."""
    vp, vs, rho, h, qsinv, per = map(_f32, (vp, vs, rho, h, qsinv, periods))
    P = per.size
    c = np.zeros(P, np.float32); u = np.zeros(P, np.float32)
    ns = ctypes.c_int(0); nd = ctypes.c_long(0)
    st = lib().surfdisp_oracle_forward(vp.size, int(kind), _fp(vp), _fp(vs), _fp(rho), _fp(h),
                                       _fp(qsinv), _fp(per), P, _fp(c), _fp(u),
                                       ctypes.byref(ns), ctypes.byref(nd))
    return dict(c=c, u=u, status=st, nsolved=ns.value, n_delta=nd.value)


def fast_surf(nlay, ilvry, vp, vs, rho, h, qsinv, per, nper):
    """f2py-shaped call (fast_surf.pyf:6-19) on the C oracle."""
    r = forward(np.asarray(vp)[:nlay], np.asarray(vs)[:nlay], np.asarray(rho)[:nlay],
                np.asarray(h)[:nlay], np.asarray(qsinv)[:nlay], np.asarray(per)[:nper], ilvry)
    outs = [np.zeros(200, np.float32) for _ in range(4)]
    ns = r["nsolved"]
    if ilvry == 2:
        outs[0][:ns] = r["u"][:ns]; outs[2][:ns] = r["c"][:ns]
    else:
        outs[1][:ns] = r["u"][:ns]; outs[3][:ns] = r["c"][:ns]
    return tuple(outs)


def forward_batch(model, periods, kind, nlay=None, nthreads=1):
    """model float32 [B,5,L] rows (vp, vs, rho, h, qsinv) -> c[B,P], u[B,P], status[B]."""
    model = np.ascontiguousarray(model, dtype=np.float32)
    B, five, Lmax = model.shape
    assert five == 5
    per = _f32(periods); P = per.size
    c = np.zeros((B, P), np.float32); u = np.zeros((B, P), np.float32)
    status = np.zeros(B, np.int32)
    nl = None
    if nlay is not None:
        nlay = np.ascontiguousarray(nlay, dtype=np.int32)
        nl = nlay.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    lib().surfdisp_oracle_forward_batch(B, Lmax, nl, _fp(model), P, _fp(per), int(kind),
                                        _fp(c), _fp(u),
                                        status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                        int(nthreads))
    return c, u, status


def partials(vp, vs, rho, h, qsinv, period, kind):
    """The analytic partials of one period as the reference's REIGEN / LEIGEN leave them in COMMON /rar1/
    (oracle/refso.py::last_partials is the same read-out of the reference itself): dict(dcda, dcdb, dcdr, dwx
    float64[1000], mmax, ndiv, c, u, status)."""
    vp, vs, rho, h, qsinv = map(_f32, (vp, vs, rho, h, qsinv))
    out = [np.zeros(1000, np.float64) for _ in range(4)]
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    mm = ctypes.c_int(0); nd = ctypes.c_int(0); c = ctypes.c_float(0); u = ctypes.c_float(0)
    st = lib().surfdisp_oracle_partials(vp.size, int(kind), _fp(vp), _fp(vs), _fp(rho), _fp(h), _fp(qsinv),
                                        float(period), dp(out[0]), dp(out[1]), dp(out[2]), dp(out[3]),
                                        ctypes.byref(mm), ctypes.byref(nd), ctypes.byref(c), ctypes.byref(u))
    return dict(dcda=out[0], dcdb=out[1], dcdr=out[2], dwx=out[3], mmax=mm.value, ndiv=nd.value,
                c=c.value, u=u.value, status=st)


def sum_sublayers(arr, nlay, ndiv, mmax, water):
    """Per-sublayer entries of COMMON /rar1/ -> per INPUT layer (sum over a layer's ndiv sublayers, surfa.f:789-820:
    layers jj..n-1 are split, jj = 2 with a water top, the half space is the last entry; the block is shifted one
    entry down when the top is solid).  Entries at or beyond ``mmax`` (layers dropped by that call) are zero."""
    a = np.asarray(arr, np.float64)
    sh = 0 if water else 1                       # entry of sublayer 1 (0-based index)
    out = np.zeros(nlay)
    jj = 2 if (water and ndiv > 1) else 1
    pos = sh
    for lay in range(1, nlay + 1):               # 1-based input layer
        nsub = 1 if (lay < jj or lay == nlay or ndiv <= 1) else ndiv
        out[lay - 1] = a[pos:pos + nsub].sum()
        pos += nsub
    return out


def group_at(model, periods, kind, c_at, nlay=None):
    """U[B,P] of the oracle's group-velocity computation (REIGEN / LEIGEN, with the ellipticity of calcul.f:195)
    evaluated at the GIVEN phase velocities c_at[B,P] (entries <= 0: at the oracle's own root); also returns the
    oracle's own c[B,P].  Test hook, see surfdisp_oracle_forward_at."""
    model = np.ascontiguousarray(model, dtype=np.float32)
    B, _, Lmax = model.shape
    per = _f32(periods); P = per.size
    c_at = np.ascontiguousarray(c_at, dtype=np.float32)
    c = np.zeros((B, P), np.float32); u = np.zeros((B, P), np.float32)
    for i in range(B):
        n = int(nlay[i]) if nlay is not None else Lmax
        m = np.ascontiguousarray(model[i, :, :n])
        lib().surfdisp_oracle_forward_at(n, int(kind), _fp(m[0]), _fp(m[1]), _fp(m[2]), _fp(m[3]), _fp(m[4]),
                                         _fp(per), P, _fp(c_at[i]), _fp(c[i]), _fp(u[i]))
    return c, u
