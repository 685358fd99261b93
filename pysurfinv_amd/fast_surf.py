"""Drop-in for the reference's f2py module ``pySurfInv.fast_surf``.

    (ur0, ul0, cr0, cl0) = fast_surf.fast_surf(nlay, ilvry, Vp, Vs, rho, h, qsinv, per, nper)

Same name, argument order and meaning, dtypes and error behaviour as the f2py
wrapper generated from fast_surf_src/fast_surf.pyf:6-19 (called at models.py:27
and senskernel.py:188):

* inputs are converted to float32 copies (f2py does the same for real*4 dummies);
  the five layer arrays must have exactly ``nlay`` elements (``depend(n_layer0)``);
  ``per`` has 200 elements of which the first ``nper`` are used;
* four NEW zero-filled float32[200] arrays are returned; only the pair that matches
  ``ilvry`` (1 Love / 2 Rayleigh) is written, and only for the solved periods
  (fast_surf.f:197-208); no exception is raised for an unsolvable stack -- the
  caller tests ``cr0[:nper] < 0.01`` (models.py:29-33).

The work runs on the MI355X through the C-ABI symbol ``fast_surf_`` of
libsurfdisp_hip.so (one stack per call: this entry exists for compatibility;
throughput comes from pysurfinv_amd.forward.forward_batch).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

__all__ = ["fast_surf"]


def _f32(a, n, name):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32)).ravel()
    if a.size != n:
        # f2py: "0-th dimension must be fixed to n but got m"
        raise ValueError(f"fast_surf: {name} must have exactly {n} elements, got {a.size}")
    return a


def fast_surf(n_layer0, kind0, a_ref0, b_ref0, rho_ref0, d_ref0, qs_ref0, cvper, ncvper):
    L = _lib.lib()
    n = int(n_layer0)
    vp = _f32(a_ref0, n, "a_ref0"); vs = _f32(b_ref0, n, "b_ref0"); rho = _f32(rho_ref0, n, "rho_ref0")
    h = _f32(d_ref0, n, "d_ref0"); qs = _f32(qs_ref0, n, "qs_ref0")
    per = _f32(cvper, _lib.NPER_MAX, "cvper")
    outs = [np.zeros(_lib.NPER_MAX, np.float32) for _ in range(4)]
    fp = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    n_, k_, p_ = ctypes.c_int(n), ctypes.c_int(int(kind0)), ctypes.c_int(int(ncvper))
    if _lib.lib().surfdisp_device_count() < 1:
        raise _lib.SurfdispError("no HIP device visible: pysurfinv_amd has no CPU fallback")
    L.fast_surf_(ctypes.byref(n_), ctypes.byref(k_), fp(vp), fp(vs), fp(rho), fp(h), fp(qs),
                 fp(per), ctypes.byref(p_), fp(outs[0]), fp(outs[1]), fp(outs[2]), fp(outs[3]))
    return tuple(outs)
