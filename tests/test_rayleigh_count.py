"""CPU, float64: the Rayleigh mode count the opt-in count-guided scan carries (csrc/surfdisp_kernels.hip, ray_step<.., CERT>;
profiles/r04b/rayleigh_count_ww.txt).  scripts/analysis/rayleigh_count_ww_state.py ports the production recursion's layer step line by
line and forms, per layer, the Wittrick-Williams count from its variables; here a small sample of its check runs in the suite:
  * per layer, the count equals the number of zeros of det U_s inside the layer found by fine stepping of the expm-propagated
    surface pair, wherever the layer's S phase is below pi (two-zero layers included);
  * the total (layers + boundary index at the half space) equals the brute-force count of the secular function's sign changes
    below the trial velocity on every such trial."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts", "analysis"))


def test_in_layer_count_and_total_count_float64():
    import rayleigh_count as rc
    import rayleigh_count_ww_state as W
    rc.rng = np.random.default_rng(5)
    npairs = nsafe = 0
    for case in range(3):
        st = rc.random_stack()
        if case == 1:                                           # soft top, one thick layer: what broke the interface-only count
            st[1][0] = 0.6; st[0][0] = 1.7; st[3][0] = 2.0; st[3][2] = 30.0
        T = (8.0, 14.0, 25.0)[case]
        cs = np.arange(0.75 * st[1][:-1].min(), st[1][-1] * 0.999, 0.02)
        R = [W.trial(st, T, float(c), nsub=48) for c in cs]
        dlt = np.array([r[0] for r in R]); N = np.array([r[2] for r in R]); safe = np.array([r[3] for r in R])
        for r in R:
            for nz, cnt, sph in r[1]:
                if sph < np.pi:
                    assert nz == cnt
                    npairs += 1
        # brute force on a finer grid: roots below each coarse trial
        cf = np.arange(cs[0], cs[-1] + 1e-9, 0.002)
        df = np.array([W.trial(st, T, float(c), nsub=1)[0] for c in cf])
        roots = cf[1:][np.sign(df[1:]) != np.sign(df[:-1])]
        brute = np.array([(roots <= c + 1e-12).sum() for c in cs])
        off = (N - brute)[safe]
        assert safe.sum() > 10 and (off == off[0]).all() and off[0] == 0
        nsafe += int(safe.sum())
    assert npairs > 500 and nsafe > 60


def _count(W, st, T, c):
    return W.trial(st, T, float(c), nsub=1)[2:4]               # (N, safe)


def test_energy_velocity_window_float64():
    """The certificate that was worked out but not built (profiles/r04b/rayleigh_count_ww.txt, item 5): at fixed k the count is
    monotone in omega and no branch is steeper than alpha_max, so if the count does not change along omega -+ alpha_max (k1 - k2) / 2
    at both coarse trials, no branch crosses omega = const between them.  Here, in float64: on the stack the count-guided scan
    failed on (tests/golden/ref_zgv_stack.npz, T = 23.78 s, coarse trials 0.926 / 0.986 km/s) the window test REFUSES; on ordinary
    stacks it accepts intervals well below the first root - and those hold no root."""
    import rayleigh_count as rc
    import rayleigh_count_ww_state as W

    def window_ok(st, T, c1, c2):
        amax = float(np.max(st[0]))
        e = 0.5 * amax * (1.0 / c1 - 1.0 / c2)
        for c in (c1, c2):
            (np_, s1), (nm, s2) = _count(W, st, T / (1 + e), c * (1 + e)), _count(W, st, T / (1 - e), c * (1 - e))
            if not (s1 and s2) or np_ != nm:
                return False
        return True

    d = np.load(os.path.join(HERE, "golden", "ref_zgv_stack.npz"))
    m = d["model"].astype(float)
    st = (m[0], m[1], m[2], m[3])
    assert not window_ok(st, float(d["periods"][4]), 0.926, 0.986)
    rc.rng = np.random.default_rng(9)
    accepted = 0
    for case in range(4):
        st = rc.random_stack(); T = float(rc.rng.uniform(10, 40))
        cs = np.arange(0.8 * st[1][:-1].min(), st[1][-1] * 0.999, 0.002)
        dl = np.array([W.trial(st, T, float(c), nsub=1)[0] for c in cs])
        roots = cs[1:][np.sign(dl[1:]) != np.sign(dl[:-1])]
        for c1 in np.arange(cs[0] + 0.01, cs[-1] - 0.07, 0.06):
            c2 = c1 + 0.06
            if window_ok(st, T, c1, c2):
                accepted += 1
                assert not ((roots > c1) & (roots < c2)).any()
    assert accepted >= 5
