#!/usr/bin/env python3
"""float64 check of the IN-LAYER zero count the interface-only evaluation lacks (profiles/r04a/rayleigh_count.txt, item 2):
Wittrick-Williams.  Surface pair integrated down; at the top of a layer its impedance is Z_t = T U^-1 (tractions ordered (tr, tz)).
With the layer's propagator P(d) in blocks [[Puu, Put], [Ptu, Ptt]], det U at depth z below the layer's top vanishes where
det(Puu(z) + Put(z) Z_t) = 0, i.e. where M(z) = Z_t - K11(z), K11 = -Put^-1 Puu (the layer clamped at z, seen from its top), is
singular.  M(0+) = +infinity (positive definite), and as long as no clamped-clamped mode of the slab lies below the frequency
(J = 0: guaranteed by Korn's inequality while the S phase k z r_beta < pi) the eigenvalues of M only move downwards:
      #{zeros of det U inside the layer} = #{negative eigenvalues of M(d)}   (0, 1 or 2).
Checked against the finely stepped count per layer."""
import sys
import numpy as np
from scipy.linalg import expm
import rayleigh_count as rc
from rayleigh_count import layer_matrix, halfspace_start

TR = [3, 2]                                   # traction rows in the order (tr, tz)


def layer_counts(stack, T, c, nsub=96):
    a, b, rho, d = stack
    om = 2 * np.pi / T; k = om / c
    Y = np.array([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0], [0.0, 0.0]])
    out = []
    for i in range(len(d) - 1):
        A = layer_matrix(k, om, a[i], b[i], rho[i])
        U, Tt = Y[:2], Y[TR]
        detU = np.linalg.det(U)
        # fine count
        Pf = expm(A * d[i] / nsub); Yf = Y.copy(); s_prev = np.sign(detU); nz = 0
        for _ in range(nsub):
            Yf = Pf @ Yf
            s = np.sign(np.linalg.det(Yf[:2]))
            if s != 0 and s != s_prev: nz += 1; s_prev = s
        P = expm(A * d[i])
        Puu = P[np.ix_([0, 1], [0, 1])]; Put = P[np.ix_([0, 1], TR)]
        sphase = k * d[i] * np.sqrt(max((c / b[i]) ** 2 - 1, 0.0))
        if abs(detU) > 1e-300 and abs(np.linalg.det(Put)) > 1e-300:
            Z = Tt @ np.linalg.inv(U)
            K11 = -np.linalg.inv(Put) @ Puu
            M = Z - K11
            ev = np.linalg.eigvalsh(0.5 * (M + M.T))
            out.append((i, nz, int((ev < 0).sum()), sphase, np.abs(M - M.T).max() / (np.abs(M).max() + 1e-300)))
        Y = P @ Y
        q, r = np.linalg.qr(Y)
        if np.linalg.det(r) < 0: q[:, 0] = -q[:, 0]
        Y = q
    return out


def main():
    rc.rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    tab = {}
    for case in range(ncase):
        st = rc.random_stack()
        if rc.rng.random() < 0.4:                         # soft sediments / thick layers: the cases that broke the interface-only count
            st[1][0] = rc.rng.uniform(0.3, 1.2); st[0][0] = max(1.7 * st[1][0], 1.6); st[3][0] = rc.rng.uniform(0.5, 4.0)
            st[3][2] = rc.rng.uniform(20, 40)
        T = float(rc.rng.uniform(3, 40))
        cmin, cmax = 0.75 * st[1][:-1].min(), st[1][-1] * 0.999
        for c in np.arange(cmin, cmax, 0.004):
            for (i, nz, nneg, sph, asym) in layer_counts(st, T, c):
                key = ("S phase < pi" if sph < np.pi else "S phase >= pi", nz, nneg)
                tab[key] = tab.get(key, 0) + 1
    print("(regime, zeros of det U inside the layer by fine stepping, negative eigenvalues of Z_t - K11): layers x trials")
    for kq in sorted(tab): print("  ", kq, tab[kq])
    bad = sum(v for (r, nz, nn), v in tab.items() if r == "S phase < pi" and nz != nn)
    tot = sum(v for (r, nz, nn), v in tab.items() if r == "S phase < pi")
    print(f"S phase < pi: rule fails in {bad} of {tot}")


if __name__ == "__main__":
    main()
