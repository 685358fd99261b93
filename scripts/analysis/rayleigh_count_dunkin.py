#!/usr/bin/env python3
"""float64 check of the Rayleigh root count in the form the production recursion could carry (companion of rayleigh_count.py).

The secular function of the reference (DLTAR4, surfa.f:193-357) propagates the five 2 x 2 minors of the two solutions that satisfy
the FREE-SURFACE condition DOWN through the stack (start: det U = 1, tractions 0) and closes with the half space's decaying pair.
Count rule to verify (Morse index of the Neumann problem + boundary index at the half space):
      N(c) = #{sign changes of det U_s at the layer interfaces, top -> bottom}
           + #{positive eigenvalues of Z_h - Z_s at the top of the half space}  (+ a constant),
Z_s = T_s U_s^-1 of the surface pair, Z_h = T_h U_h^-1 of the half space's decaying pair, tractions ordered (tr, tz).
Everything in it is a ratio of the minors the recursion already carries.  Also reports how thick (in S phase, k d r_beta) a layer
may be before the interface-only count misses zeros of det U inside it."""
import sys
import numpy as np
from scipy.linalg import expm
from rayleigh_count import layer_matrix, halfspace_start, random_stack, rng as _r
import rayleigh_count as rc


def sweep_down(stack, T, c, nsub):
    a, b, rho, d = stack
    om = 2 * np.pi / T; k = om / c
    Y = np.array([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0], [0.0, 0.0]])           # U = I, T = 0 at the free surface
    s_prev = 1.0; nz = 0
    for i in range(len(d) - 1):
        P = expm(layer_matrix(k, om, a[i], b[i], rho[i]) * d[i] / nsub)       # downwards: + dz
        for _ in range(nsub):
            Y = P @ Y
            s = np.sign(np.linalg.det(Y[:2]))
            if s != 0 and s != s_prev: nz += 1; s_prev = s
        q, r = np.linalg.qr(Y)
        if np.linalg.det(r) < 0: q[:, 0] = -q[:, 0]
        Y = q
        s_prev = np.sign(np.linalg.det(Y[:2]))
    Yh = halfspace_start(k, om, c, a[-1], b[-1], rho[-1])
    Zs = Y[[3, 2]] @ np.linalg.inv(Y[:2]); Zh = Yh[[3, 2]] @ np.linalg.inv(Yh[:2])
    S = Zh - Zs
    ev = np.linalg.eigvalsh(0.5 * (S + S.T))
    delta = np.linalg.det(np.hstack([Y, Yh]))
    return delta, nz, int((ev > 0).sum()), int((ev < 0).sum())


def main():
    rc.rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    hold = {"flips + pos": 0, "flips + neg": 0}
    thick = []
    for case in range(ncase):
        st = rc.random_stack(); T = float(rc.rng.uniform(4, 40))
        cmin, cmax = 0.75 * st[1][:-1].min(), st[1][-1] * 0.999
        cs = np.arange(cmin, cmax, 0.002)
        R = np.array([sweep_down(st, T, c, 48) for c in cs])
        dlt, nz, npos, nneg = R[:, 0], R[:, 1], R[:, 2], R[:, 3]
        brute = np.concatenate([[0], np.cumsum(np.sign(dlt[1:]) != np.sign(dlt[:-1]))])
        for name, cnt in (("flips + pos", nz + npos), ("flips + neg", nz + nneg)):
            off = cnt - brute
            hold[name] += int((off == off[0]).all())
        # interface-only flips (one step per layer) against the finely stepped count, and the largest S phase of a layer at that c
        om = 2 * np.pi / T
        for c in cs[:: max(1, len(cs) // 40)]:
            one = sweep_down(st, T, c, 1)[1]; fine = sweep_down(st, T, c, 48)[1]
            ph = max(om / c * st[3][i] * np.sqrt(max((c / st[1][i]) ** 2 - 1, 0.0)) for i in range(len(st[3]) - 1))
            thick.append((ph, one == fine))
        print(f"case {case}: L={len(st[3])} T={T:.1f} modes {brute[-1]}  offsets flips+pos {np.unique(nz + npos - brute)}  flips+neg {np.unique(nz + nneg - brute)}", flush=True)
    print("rule holds (constant offset over the whole range of c) in", hold, "of", ncase, "cases")
    thick = np.array(thick)
    for lo, hi in ((0, 0.5), (0.5, 1.0), (1.0, 1.57), (1.57, 2.2), (2.2, 3.14), (3.14, 99)):
        sel = (thick[:, 0] >= lo) & (thick[:, 0] < hi)
        if sel.any():
            print(f"largest S phase of a layer in [{lo}, {hi}) rad: interface-only count equals the fine count in {thick[sel, 1].mean():.3f} of {sel.sum()} trials")


if __name__ == "__main__":
    main()
