#!/usr/bin/env python3
"""Parses the reference's own sensitivity-kernel fixtures (data files of its known-answer test,
``senskernel-1.0/TEST1/test.{phv,grv}.{R,L}_0_{T}``, fundamental mode, T = 10..100 s) into
tests/golden/test1_kernels.npz.  Columns (KERNELS.csh:80-92): depth km, (dc/c)/(db/b),
[(dc/c)/(da/a) Rayleigh only,] (dc/c)/(drho/rho), all per km; the first row also carries
period, c, U, mode.

    python tests/golden/make_golden_kernels.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
T1 = "/root/reference/senskernel-1.0/TEST1"
NZ = 200


def main():
    out = {}
    periods = list(range(10, 101, 10))
    for w, ncol in (("R", 4), ("L", 3)):
        for y in ("phv", "grv"):
            K, hdr = [], []
            for T in periods:
                rows = [ln.split() for ln in open(os.path.join(T1, f"test.{y}.{w}_0_{T}")) if ln.split()]
                hdr.append([float(x) for x in rows[0][ncol:ncol + 3]])          # period, c, U
                k = np.full((NZ, ncol), np.nan)                                 # 2 km grid, 0..398 km
                k[:, 0] = 2.0 * np.arange(NZ)
                a = np.array([[float(x) for x in r[:ncol]] for r in rows])[:NZ]
                assert np.allclose(a[:, 0], k[:len(a), 0])
                k[:len(a), 1:] = a[:, 1:]
                K.append(k)
            K = np.array(K)                                                     # [P, nz, ncol]
            out[f"{y}_{w}_depth"] = K[0, :, 0]
            out[f"{y}_{w}_kernels"] = K[:, :, 1:]
            out[f"{y}_{w}_header"] = np.array(hdr)
            print(y, w, K.shape, hdr[1])
    out["periods"] = np.array(periods, float)
    np.savez_compressed(os.path.join(HERE, "test1_kernels.npz"), **out)


if __name__ == "__main__":
    main()
