"""ctypes loader for oracle/_ref/libfast_surf_ref.so  --  TEST INFRASTRUCTURE ONLY.

The shared object is the *unmodified reference Fortran* (fast_surf_src/*.f) built
by oracle/build_ref.sh with AMD flang.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; it is the checker / reported
CPU baseline, never a product path.

Calling convention = what f2py generates from fast_surf_src/fast_surf.pyf:6-19:
all scalars by reference, float32 arrays, four zero-filled float32[200] outputs.

"Fresh-process" semantics (defect 1): the Fortran keeps
``ndiv`` in COMMON /c/ and clamps it in place (surfa.f:414-415, :783-784), so a
solve depends on the history of the process.  ``fast_surf(..., fresh=True)``
(the default) resets ndiv=5 (init.f:25 DATA value) and zeroes COMMON /dispe/
(stale-output defect, calcul.f:173-189) before every call, which is exactly what
a new process would see.
"""
from __future__ import annotations

import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libfast_surf_ref.so")
_lib = None

NPER_MAX = 200  # fast_surf.f:9 parameter nper=200


def available() -> bool:
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise FileNotFoundError(
                f"{_SO} missing: run oracle/build_ref.sh where /root/reference exists")
        _lib = ctypes.CDLL(_SO)
    return _lib


def _reset_state(L):
    # COMMON /c/ nmax,mmax,kmax,idrop,iedit,ndiv,mode,fact,ra_1  (fast_surf.f:51)
    c_blk = (ctypes.c_int32 * 9).in_dll(L, "c_")
    c_blk[5] = 5
    # COMMON /dispe/ per_R(200),per_L(200),uR(200,2),uL(200,2),cR(200,2),cL(200,2)
    dispe = (ctypes.c_float * (200 * 10)).in_dll(L, "dispe_")
    ctypes.memset(dispe, 0, ctypes.sizeof(dispe))


def last_ratio(nper):
    """COMMON /o/ ratio(k, 1), k = 1..nper, as the last call left it: the Rayleigh ellipticity of every period
    (calcul.f:195).  CALCUL is the routine that writes it and declares the block with nper = 2000 (calcul.f:5,7:
    c(2000,20), t(2000), ratio(2000,20)) - fast_surf.f declares 200 - so the value of period k sits at float
    2000 * 20 + 2000 + (k - 1) of the symbol ``o_``."""
    o = (ctypes.c_float * (2000 * 20 + 2000 + 2000 * 20)).in_dll(lib(), "o_")
    base = 2000 * 20 + 2000
    return np.array([o[base + k] for k in range(int(nper))], np.float32)


def last_partials(nsub=None):
    """COMMON /rar1/ dcda, dcdb, dcdr, dwx (real*8[1000] each; surfa.f:390,396 / 722,729) and COMMON /rar/ mmax as the
    LAST REIGEN / LEIGEN call left them: the analytic partial derivatives of the phase velocity with respect to the
    P velocity, S velocity and density of every SUBLAYER of the flattened, attenuation-corrected stack that call
    worked on (surfa.f:1133-1135, 1182-1184, 1204-1207; Love 511-512, 564-565, 582-583), shifted one entry down
    when the top layer is solid (entry 1 = the surface, zero: surfa.f:1211-1249 / 608-631).  CALCUL calls the
    routine once per period, so make ONE-period calls to read a given period's values.  Returns a dict with the
    four arrays (first ``nsub`` entries, default all 1000), ``mmax`` (COMMON /rar/, after the shift) and ``ndiv``
    (COMMON /c/, as clamped by the call: surfa.f:783-784 / 414-415)."""
    L = lib()
    blk = (ctypes.c_double * 4000).in_dll(L, "rar1_")
    a = np.frombuffer(blk, dtype=np.float64).copy().reshape(4, 1000)
    # COMMON /rar/ dept1(1000), ampur(1000), ampuz(1000), stresz(1000), stresr(1000), mmax
    rar = (ctypes.c_int32 * 5001).in_dll(L, "rar_")
    c_blk = (ctypes.c_int32 * 9).in_dll(L, "c_")
    n = 1000 if nsub is None else int(nsub)
    return dict(dcda=a[0, :n], dcdb=a[1, :n], dcdr=a[2, :n], dwx=a[3, :n], mmax=int(rar[5000]), ndiv=int(c_blk[5]))


def fast_surf(nlay, ilvry, vp, vs, rho, h, qsinv, per, nper, fresh=True):
    """Reference fast_surf(): returns (ur0, ul0, cr0, cl0), float32[200] each."""
    L = lib()
    if fresh:
        _reset_state(L)
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))
    vp, vs, rho, h, qsinv = map(f32, (vp, vs, rho, h, qsinv))
    for a in (vp, vs, rho, h, qsinv):
        if a.size != nlay:
            raise ValueError("layer arrays must have exactly nlay elements")
    per200 = np.zeros(NPER_MAX, np.float32)
    p = np.asarray(per, dtype=np.float64).astype(np.float32)
    per200[: p.size] = p[:NPER_MAX]
    outs = [np.zeros(NPER_MAX, np.float32) for _ in range(4)]
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    n_ = ctypes.c_int(int(nlay))
    k_ = ctypes.c_int(int(ilvry))
    np_ = ctypes.c_int(int(nper))
    L.fast_surf_(ctypes.byref(n_), ctypes.byref(k_), fp(vp), fp(vs), fp(rho), fp(h),
                 fp(qsinv), fp(per200), ctypes.byref(np_),
                 fp(outs[0]), fp(outs[1]), fp(outs[2]), fp(outs[3]))
    return tuple(outs)


def forward_batch(vp, vs, rho, h, qsinv, periods, kind, want_ratio=False):
    """Loop of reference calls over a [B, L] batch -> (c[B,P], u[B,P]) float32 (+ the ellipticities ratio[B,P])."""
    vp = np.asarray(vp); B, Ln = vp.shape
    P = len(periods)
    c = np.zeros((B, P), np.float32); u = np.zeros((B, P), np.float32); r = np.zeros((B, P), np.float32)
    for i in range(B):
        ur, ul, cr, cl = fast_surf(Ln, kind, vp[i], vs[i], rho[i], h[i], qsinv[i], periods, P)
        if kind == 2:
            c[i], u[i] = cr[:P], ur[:P]
        else:
            c[i], u[i] = cl[:P], ul[:P]
        if want_ratio:
            r[i] = last_ratio(P)
    return (c, u, r) if want_ratio else (c, u)
