"""Development-container helper: import the reference Python package (read-only, where it lies)
with stub modules standing in for its un-vendored dependencies, so that golden vectors can be
captured from it.  Used only by tests/golden/make_golden_driver.py; never on the GPU box.

SURVEY.md section 8(c) recipe: symlink ``pySurfInv -> /root/reference`` in a scratch dir; stub
``Triforce.*``, ``netCDF4``, ``geographiclib``; route ``pySurfInv.fast_surf`` to the flang build of
the reference Fortran (oracle/_ref).
"""
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def install(reference_root="/root/reference"):
    sys.dont_write_bytecode = True
    scratch = tempfile.mkdtemp(prefix="refenv_")
    os.symlink(reference_root, os.path.join(scratch, "pySurfInv"))
    sys.path.insert(0, scratch)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Dummy:
        def __getattr__(self, k):
            return Dummy()

        def __call__(self, *a, **k):
            return Dummy()

    mod("Triforce").__path__ = []
    mod("Triforce.pltHead", plt=Dummy(), mpl=Dummy(), np=np)
    mod("Triforce.obspyPlus", randString=lambda n: "X" * n)
    # gaussFun(A, mu, sig, x): un-vendored; the standard three-parameter Gaussian stands in (the form
    # pysurfinv_amd.layers_batch assumes for the Crust 'Gauss' option, layers.py:176-183) - parity of that one
    # function's exact form is unpinned, the code path around it is the reference's
    mod("Triforce.mathPlus", logQuad=Dummy(), gaussFun=lambda A, mu, sig, x: A * np.exp(-(np.asarray(x) - mu) ** 2 / (2.0 * sig ** 2)))
    mod("Triforce.utils", GeoGrid=Dummy(), GeoMap=Dummy())
    mod("Triforce.customPlot", addAxes=Dummy(), addCAxes=Dummy())
    mod("netCDF4", Dataset=Dummy())
    mod("geographiclib").__path__ = []
    mod("geographiclib.geodesic", Geodesic=Dummy())
    from oracle import refso
    mod("pySurfInv.fast_surf", fast_surf=lambda *a: refso.fast_surf(*a))
    return scratch
