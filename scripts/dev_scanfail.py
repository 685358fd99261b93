#!/usr/bin/env python3
"""Development (CPU): stacks the scan soak saved because default and exact scan disagreed -> the oracle's
secular function on the fine grid around the first differing period."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsurfdisp_oracle.so"))
fp = ctypes.POINTER(ctypes.c_float); ip = ctypes.POINTER(ctypes.c_int)
S = np.load(os.path.join(ROOT, "gpurun_out", os.environ.get("SCANFAIL", "scanfail.npy")), allow_pickle=True)
nshow = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for idx, s in enumerate(S[:nshow]):
    m = np.ascontiguousarray(s["model"], np.float32); per = np.ascontiguousarray(s["per"], np.float32)
    ce, cd = s["c_exact"], s["c_default"]
    k = int(np.nonzero(ce != cd)[0][0])
    n = m.shape[1]
    cap = 4000
    ct = np.zeros(cap, np.float32); dt = np.zeros(cap, np.float32); mt = np.zeros(cap, np.int32)
    a = [np.ascontiguousarray(m[i]) for i in range(5)]
    cnt = L.surfdisp_oracle_scan_trace(n, int(s["kind"]), *[x.ctypes.data_as(fp) for x in a], per.ctypes.data_as(fp), len(per),
                                       k, 60, ct.ctypes.data_as(fp), dt.ctypes.data_as(fp), mt.ctypes.data_as(ip), cap)
    print(f"--- case {idx}: kind {s['kind']} team {s['team']} L={n} first differing period k={k} T={per[k]:.3f}  exact c={ce[k]:.5f} default c={cd[k]:.5f}  (previous c {ce[k-1] if k else 0:.5f})")
    print("    vs ", np.array2string(m[1][:8], precision=3), " vp ", np.array2string(m[0][:8], precision=3), " h ", np.array2string(m[3][:8], precision=3))
    lo = max(0, cnt - 60 - 24)
    for i in range(lo, cnt):
        print(f"      c={ct[i]:.5f}  D={dt[i]: .5e}  mm={mt[i]}")

    # replay the certificate on the oracle's values: which coarse intervals holding a fine-grid sign change pass?
    G = int(s["team"]); ST = 4
    sg = lambda x: (np.signbit(x) and not np.isnan(x))
    pts = [G - 1 + ST * i for i in range(0, (cnt - G) // ST)]        # coarse grid indices (p0 = last fine point of pass 1)
    for a in range(1, len(pts)):
        i0, i1 = pts[a - 1], pts[a]
        has = any(sg(dt[i]) != sg(dt[i + 1]) for i in range(i0, i1))
        if not has or sg(dt[i0]) != sg(dt[i1]):
            continue
        j = (a - 1) % G                                              # lane of the interval's right end
        pd, val = dt[i0], dt[i1]
        lim = 2 * min(abs(pd), abs(val))
        okf = True
        if j < G - 1 and a + 1 < len(pts):
            nx = dt[pts[a + 1]]
            okf = mt[i0] == mt[i1] == mt[pts[a + 1]] and np.isfinite(nx) and abs(pd - 2 * val + nx) < lim
        okb = True
        if a >= 2:
            pp = dt[pts[a - 2]]
            okb = mt[pts[a - 2]] == mt[i0] == mt[i1] and np.isfinite(pp) and abs(pp - 2 * pd + val) < lim
        print(f"    coarse interval ({ct[i0]:.5f},{ct[i1]:.5f}) lane {j} hides a root pair: okf={okf} okb={okb}   D: {dt[i0]:.3e} .. {[float(f'{x:.3e}') for x in dt[i0+1:i1]]} .. {dt[i1]:.3e}")
