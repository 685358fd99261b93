#!/usr/bin/env python3
"""tests/golden/ref_zgv_stack.npz: the one stack of 4.95e8 on which the opt-in count-guided Rayleigh scan (SURFDISP_FASTSCAN) left the
point-by-point scan (scripts/soak_cert.py SOAK_KIND=2, dump gpurun_out/rcert_ww_dump.npz) - a soft-sediment stack whose secular
function dips below zero between two coarse trials on a branch with a zero-group-velocity point (profiles/r04b/rayleigh_count_ww.txt).
Reference outputs from the unmodified Fortran (oracle/_ref); `c_count_guided` = what the count-guided scan returned.
usage: python tests/golden/make_golden_zgv.py gpurun_out/rcert_ww_dump.npz"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import refso
d = np.load(sys.argv[1])
n = int(d["nlay"][0]); P = int(d["P"][0])
m = np.ascontiguousarray(d["model"][0][:, :n], dtype=np.float32); per = np.ascontiguousarray(d["per"][0][:P], dtype=np.float32)
r = refso.fast_surf(n, 2, m[0], m[1], m[2], m[3], m[4], per, P)
np.savez_compressed(os.path.join(HERE, "ref_zgv_stack.npz"), model=m, periods=per, c=np.asarray(r[2][:P], np.float32),
                    u=np.asarray(r[0][:P], np.float32), c_count_guided=d["c1"][0][:P])
