#!/usr/bin/env python3
"""float64 check of the in-layer count (rayleigh_count_ww.py) IN THE PRODUCTION RECURSION'S OWN VARIABLES (ray_step of
csrc/surfdisp_kernels.hip, ported line by line): state (b1, h2..h5) at the top of a layer, n1 = the new b1,
      c5 = rsinp rsinq + sinpr sinqr + 2 (1 - cosp cosq)      (coefficient of h5 in n1: det Put up to a positive factor)
      c2 = -(rsinp cosq + cosp sinqr)                         (coefficient of h2)
      det M ~ n1 / (b1 c5),   M11 ~ (h4 c5 - c2 b1) / (b1 c5)
      zeros of det U inside the layer = 1 if det M < 0, else 2 if M11 < 0, else 0        (S phase of the layer < pi)
against the finely stepped count of the expm-propagated surface pair, and the total
      N(c) = sum over the layers + #{positive eigenvalues of Z_h - Z_s} against the brute-force count of roots below c."""
import sys
import numpy as np
from scipy.linalg import expm
import rayleigh_count as rc
from rayleigh_count import layer_matrix, halfspace_start

TR = [3, 2]


def ray_step(s, wvno, csq, sv, d, a, rho_prev, rho, first):
    b1, h2, h3, h4, h5 = s
    if not first:
        rat = rho_prev / rho
        h2 *= rat; h3 *= rat; h4 *= rat; h5 *= rat * rat
    icsq = 1.0 / csq
    arga = 1.0 - csq / (a * a); argb = 1.0 - csq / (sv * sv)
    xa = max(abs(arga), 1e-300); xb = max(abs(argb), 1e-300)
    ra = np.copysign(np.sqrt(xa), -arga); rb = np.copysign(np.sqrt(xb), -argb)
    wd = wvno * d; g = 2 * sv * sv * icsq; g1 = g - 1
    pm = wd * ra; qm = wd * rb
    if arga > 0: sh, ch = np.sinh(pm), np.cosh(pm); rsinp = -ra * sh; sinpr = sh / ra; cosp = ch
    else: sn, cs = np.sin(pm), np.cos(pm); rsinp = ra * sn; sinpr = sn / ra; cosp = cs
    if not argb > 0: sn, cs = np.sin(qm), np.cos(qm); rsinq = rb * sn; sinqr = sn / rb; cosq = cs
    else: sh, ch = np.sinh(qm), np.cosh(qm); rsinq = -rb * sh; sinqr = sh / rb; cosq = ch
    g2, g12 = g * g, g1 * g1
    u1 = g2 * b1 + 2 * g * h3 - h5; u2 = g12 * b1 + 2 * g1 * h3 - h5
    D = 1 - cosp * cosq
    t1 = rsinq * u1 + cosq * h2; t2 = sinqr * u2 - cosq * h4
    E1 = rsinp * t1 - cosp * rsinq * h4 + D * u2
    E2 = sinpr * t2 + cosp * sinqr * h2 + D * u1
    n1 = b1 - E1 - E2
    n3 = g * E1 + g1 * E2 + h3; n5 = g2 * E1 + g12 * E2 + h5
    n2 = cosp * t1 + sinpr * (rsinq * h4 + cosq * u2)
    n4 = rsinp * (sinqr * h2 - cosq * u1) - cosp * t2
    c5 = rsinp * rsinq + sinpr * sinqr + 2 * D
    c2 = -(rsinp * cosq + cosp * sinqr)
    detM = n1 / (b1 * c5); m11 = (h4 * c5 - c2 * b1) / (b1 * c5)
    cnt = 1 if detM < 0 else (2 if m11 < 0 else 0)
    sphase = qm if not argb > 0 else 0.0
    return (n1, n2, n3, n4, n5), cnt, sphase


def trial(stack, T, c, nsub=96):
    a, b, rho, d = stack
    om = 2 * np.pi / T; k = om / c
    Y = np.array([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0], [0.0, 0.0]])
    s = (1.0, 0.0, 0.0, 0.0, 0.0)
    rows = []; total = 0; safe = True
    for i in range(len(d) - 1):
        A = layer_matrix(k, om, a[i], b[i], rho[i])
        Pf = expm(A * d[i] / nsub); s_prev = np.sign(np.linalg.det(Y[:2])); nz = 0
        for _ in range(nsub):
            Y = Pf @ Y
            sg = np.sign(np.linalg.det(Y[:2]))
            if sg != 0 and sg != s_prev: nz += 1; s_prev = sg
        q, r = np.linalg.qr(Y)
        if np.linalg.det(r) < 0: q[:, 0] = -q[:, 0]
        Y = q
        s, cnt, sph = ray_step(s, k, c * c, b[i], d[i], a[i], rho[i - 1] if i else rho[0], rho[i], i == 0)
        mx = max(abs(v) for v in s); s = tuple(v / mx for v in s)               # (positive rescaling: signs unchanged)
        rows.append((nz, cnt, sph)); total += cnt; safe = safe and sph < np.pi
    Yh = halfspace_start(k, om, c, a[-1], b[-1], rho[-1])
    Zs = Y[TR] @ np.linalg.inv(Y[:2]); Zh = Yh[TR] @ np.linalg.inv(Yh[:2])
    ev = np.linalg.eigvalsh(0.5 * ((Zh - Zs) + (Zh - Zs).T))
    delta = np.linalg.det(np.hstack([Y, Yh]))
    return delta, rows, total + int((ev > 0).sum()), safe


def main():
    rc.rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    lay = {}; okN = totN = 0
    for case in range(ncase):
        st = rc.random_stack()
        if rc.rng.random() < 0.4:
            st[1][0] = rc.rng.uniform(0.3, 1.2); st[0][0] = max(1.7 * st[1][0], 1.6); st[3][0] = rc.rng.uniform(0.5, 4.0)
            st[3][2] = rc.rng.uniform(20, 40)
        T = float(rc.rng.uniform(3, 40))
        cs = np.arange(0.75 * st[1][:-1].min(), st[1][-1] * 0.999, 0.003)
        R = [trial(st, T, c) for c in cs]
        dlt = np.array([r[0] for r in R]); N = np.array([r[2] for r in R]); safe = np.array([r[3] for r in R])
        brute = np.concatenate([[0], np.cumsum(np.sign(dlt[1:]) != np.sign(dlt[:-1]))])
        off = (N - brute)[safe]
        okN += int((off == 0).sum()); totN += int(safe.sum())
        for r in R:
            for nz, cnt, sph in r[1]:
                key = ("S<pi" if sph < np.pi else "S>=pi", nz, cnt); lay[key] = lay.get(key, 0) + 1
        print(f"case {case}: L={len(st[3])} T={T:.1f} modes {brute[-1]} safe trials {safe.sum()}/{len(cs)} offsets among safe trials {np.unique(off)}", flush=True)
    print("(regime, fine count, rule from the recursion's variables): layers x trials")
    for kq in sorted(lay): print("  ", kq, lay[kq])
    print(f"total count N(c) = brute-force count on {okN} of {totN} safe trials")


if __name__ == "__main__":
    main()
