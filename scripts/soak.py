#!/usr/bin/env python3
"""Randomised differential soak: HIP path vs CPU oracle over random stacks / period lists / team
sizes.  Reports the worst disagreements; exits non-zero on a hang-free but wrong result.
SOAK_AGAINST=strict compares the default mode with the library's own verification mode instead (SURFDISP_STRICT: every
stack through the statement-by-statement kernel) - no CPU in the loop, so batches are 16 x larger and a minute covers
tens of millions of stacks; what it cannot see is an error the two kernels share (the oracle soak covers that)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysurfinv_amd import _lib, forward, synth
from oracle import cport

rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "0")))
STRICT = os.environ.get("SOAK_AGAINST") == "strict"
T_END = time.time() + float(os.environ.get("SOAK_SECONDS", "120"))
worst = dict(c=0.0, u=0.0)
T_LAST = time.time()
nstack = ncase = npat = novf = nbigc = nbigu = 0
bad_cases = []
# every offending STACK is kept (up to SOAK_KEEP), padded to a common shape, for tests/golden/make_golden_offenders.py
KEEP = int(os.environ.get("SOAK_KEEP", "6000")); LCAP, PCAP = 48, 40
off = dict(model=[], nlay=[], per=[], P=[], kind=[], team=[], cat=[], c=[], co=[], u=[], uo=[])
TAG = os.environ.get("SOAK_TAG", (os.environ.get("SOAK_FAMILY") or "general") + ("_strict" if STRICT else "_oracle"))
if os.environ.get("SOAK_RECLASSIFY"):                      # only the classification, of the offenders an earlier soak saved
    _f = np.load(os.environ["SOAK_RECLASSIFY"])
    off = {k: list(_f[k]) for k in off}
    T_END = 0.0
while time.time() < T_END:
    L = int(rng.integers(2, 48)); B = int(rng.integers(64, 2048)) * (16 if STRICT else 1); kind = int(rng.integers(1, 3))
    if os.environ.get("SOAK_KIND"):                        # one wave type only (1 Love, 2 Rayleigh)
        kind = int(os.environ["SOAK_KIND"])
    noise = float(rng.choice([0.02, 0.05, 0.1, 0.2])); mono = bool(rng.random() < 0.6)
    model = synth.synth_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise, monotone=mono,
                               total_thickness=float(rng.choice([60., 120., 200., 400.])))
    if rng.random() < 0.2 and L >= 4:                       # water + sediment on top
        model[:, 1, 0] = 0.0; model[:, 0, 0] = 1.475; model[:, 2, 0] = 1.027; model[:, 4, 0] = 1e-4
        model[:, 3, 0] = rng.uniform(0.3, 4.0, B)
        model[:, 1, 1] = rng.uniform(0.6, 1.8, B); model[:, 0, 1] = 1.23 * model[:, 1, 1] + 1.28
        model[:, 3, 1] = rng.uniform(0.2, 2.0, B)
    sed = os.environ.get("SOAK_FAMILY") == "sediment" and L >= 4 and rng.random() < 0.7
    if sed:                                                 # soft sediments over rock (synth.sediment_models)
        model = synth.sediment_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise,
                                      total_thickness=float(rng.choice([30., 60., 120., 200., 400.])))
    ovf = os.environ.get("SOAK_FAMILY") == "overflow"
    if ovf:                                                 # two to five layers of 60-250 km: the fp32 secular function overflows
        L = int(rng.integers(2, 6))
        model = synth.synth_models(B, L, seed=int(rng.integers(1 << 30)), noise=noise, monotone=mono,
                                   total_thickness=float(rng.uniform(60., 250.)) * L)
    nlay = None
    if rng.random() < 0.3 and L > 3:
        nlay = rng.integers(2, L + 1, B).astype(np.int32)
    P = int(rng.integers(1, 40))
    per = np.sort(rng.uniform(3.0, 150.0, P)).astype(np.float32)
    if rng.random() < 0.5:
        per = np.linspace(rng.uniform(4, 12), rng.uniform(40, 120), P).astype(np.float32)
    if sed:
        per = np.sort(rng.uniform(0.3, 30.0, P)).astype(np.float32)
    if ovf:
        per = np.sort(rng.uniform(2.5, 40.0, P)).astype(np.float32)
    team = int(rng.choice([0, 1, 2, 4, 8, 16, 32, 64]))
    _lib.lib().surfdisp_set_team(team)
    c, u, st = forward.forward_batch(model, per, kind, nlay=nlay)
    if STRICT:
        co, uo, so = forward.forward_batch(model, per, kind, nlay=nlay, strict=True)
        so = np.where(so == 8, 3, 0)                        # SURFDISP_NUMERIC <-> the oracle's NEVILL failure code
    else:
        co, uo, so = cport.forward_batch(model, per, kind, nlay=nlay, nthreads=32)
    rows = ((c > 0) == (co > 0)).all(axis=1)
    ok = (co != 0) & rows[:, None]
    ec = np.abs(c[ok] / co[ok] - 1) if ok.any() else np.zeros(1)
    eu = np.abs(u[ok] / uo[ok] - 1) if ok.any() else np.zeros(1)
    nstack += B; ncase += 1; npat += int((~rows).sum())
    # per STACK: zero pattern equal, but some phase velocity off by more than the bar ("another root"), or - with every
    # phase velocity inside the bar - some group velocity off by more than the bar
    with np.errstate(all="ignore"):
        e_c = np.where(ok, np.abs(c / np.where(co != 0, co, 1) - 1), 0.0)
        e_u = np.where(ok, np.nan_to_num(np.abs(u / np.where(uo != 0, uo, 1) - 1), nan=9.0), 0.0)
        e_u = np.where(ok & ~np.isfinite(uo) & ~np.isfinite(u), 0.0, e_u)        # NaN for NaN is agreement
    bigc = rows & (e_c.max(axis=1) > 1e-4)
    bigu = rows & ~bigc & (e_u.max(axis=1) > 1e-4)
    nbigc += int(bigc.sum()); nbigu += int(bigu.sum())
    for cat, sel in ((0, ~rows), (1, bigc), (2, bigu)):
        for i in np.nonzero(sel)[0]:
            if len(off["cat"]) >= KEEP: break
            m = np.zeros((5, LCAP), np.float32); m[:, :L] = model[i]
            pp = np.zeros(PCAP, np.float32); pp[:P] = per
            def pad(a):
                o = np.zeros(PCAP, np.float32); o[:P] = a[i]; return o
            off["model"].append(m); off["nlay"].append(int(nlay[i]) if nlay is not None else L); off["per"].append(pp)
            off["P"].append(P); off["kind"].append(kind); off["team"].append(team); off["cat"].append(cat)
            off["c"].append(pad(c)); off["co"].append(pad(co)); off["u"].append(pad(u)); off["uo"].append(pad(uo))
    novf += int((~rows & ((st == 8) | (so == 3))).sum())     # fp32-overflow regime: either side gave up
    q = np.quantile(eu, 0.999) if eu.size > 1000 else eu.max()
    if ec.max() > 2e-5 or q > 1e-4 or (~rows).mean() > 0.02 or not np.isfinite(c).all() or not np.isfinite(u).all():
        bad_cases.append((L, B, kind, noise, mono, P, team, float(ec.max()), float(eu.max()), float(q), int((~rows).sum())))
        if len(bad_cases) <= int(os.environ.get("SOAK_SAVE", "12")):
            # keep only the offending stacks (largest errors) so the file stays small
            e = np.zeros(c.shape); e[ok] = np.maximum(np.abs(c[ok] / co[ok] - 1), np.nan_to_num(np.abs(u[ok] / uo[ok] - 1), nan=9.0))
            score = e.max(axis=1) + (~rows) * 10 + (~np.isfinite(u).all(axis=1)) * 5
            idx = np.argsort(score)[-8:]
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"soak_fail_{len(bad_cases):02d}.npz"),
                                model=model[idx], per=per, kind=kind, team=team,
                                nlay=(nlay[idx] if nlay is not None else np.full(len(idx), L, np.int32)),
                                c=c[idx], u=u[idx], co=co[idx], uo=uo[idx], st=st[idx], so=so[idx])
    worst["c"] = max(worst["c"], float(ec.max())); worst["u"] = max(worst["u"], float(q))
    if time.time() - T_LAST > 45:                           # progress line (the GPU box kills silent runs)
        T_LAST = time.time()
        print(f"  ... {ncase} cases, {nstack} stacks, pattern mismatches {npat}, flagged {len(bad_cases)}", flush=True)
_lib.lib().surfdisp_set_team(0)
if off["cat"] and not os.environ.get("SOAK_RECLASSIFY"):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"soak_offenders_{TAG}.npz"),
                        **{k: np.asarray(v) for k, v in off.items()})
# ---- why: every saved offender against the oracle's conditioning and against the reference's own two builds (the criterion of
# tests/golden/u_exceptions*.json): a group velocity off by > 1e-4 is EXPLAINED when the entry is ill conditioned (the oracle's
# U moves by > 1e-4 for a 2e-6 change of c), when the reference's FMA and non-FMA builds disagree there by > 2e-5 or one of them
# returns NaN, or in the overflow regime (NaN on either side); a phase velocity off by > 1e-4 only when the reference's two builds
# disagree by that much themselves
def near_scan_point(m, n, kind, per, P, k, croot, tol=3e-6):
    """Is croot within tol (relative) of one of the trial velocities of the oracle's scan of period k (its 0.01 km/s grid)?
    Then the SIGN of the secular function at that one trial - the remainder of a cancellation to the last bit - decides which
    bracket the scan stops at: the mechanism behind every zero-pattern mismatch the soaks have shown, and, where overtones are
    0.01 km/s apart, behind a root on the neighbouring overtone."""
    import ctypes
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    cap = 8192
    ct = np.zeros(cap, np.float32); dt = np.zeros(cap, np.float32); mt = np.zeros(cap, np.int32)
    rows = [np.ascontiguousarray(m[r], dtype=np.float32) for r in range(5)]
    pp = np.ascontiguousarray(per, dtype=np.float32)
    nt = cport.lib().surfdisp_oracle_scan_trace(int(n), int(kind), *[r.ctypes.data_as(fp) for r in rows], pp.ctypes.data_as(fp), int(P), int(k), 4,
                                                ct.ctypes.data_as(fp), dt.ctypes.data_as(fp), mt.ctypes.data_as(ip), cap)
    if nt <= 0 or not croot > 0: return False
    if (np.abs(ct[:nt] / np.float32(croot) - 1) < tol).any(): return True
    # ... or a trial whose value is below 2e-3 of BOTH neighbours' (a root a few 1e-5 km/s from the trial; thick layers with
    # c ~ b: c^2/b^2 - 1 carries a relative rounding of 1e-5 there and the phase k d rb of tens of radians turns it into 1e-3 of
    # the function's scale): the same mechanism with a wider band
    a = np.abs(dt[:nt])
    with np.errstate(all="ignore"):
        inner = (a[1:-1] < 2e-3 * a[:-2]) & (a[1:-1] < 2e-3 * a[2:])
    return bool(inner.any())

def classify():
    from oracle import refso
    DC = 2e-6
    have_ref = os.path.exists(refso._SO) and os.path.exists(refso._SO.replace("_ref.so", "_ref_fma.so"))
    libs = [refso._SO, refso._SO.replace("_ref.so", "_ref_fma.so")]
    counts = {}
    unexpl = []
    for q in range(min(len(off["cat"]), int(os.environ.get("SOAK_CLASSIFY", "2500")))):
        cat = off["cat"][q]; n = off["nlay"][q]; P = off["P"][q]; kind = off["kind"][q]
        m = np.ascontiguousarray(off["model"][q][:, :n]); per = off["per"][q][:P].copy()
        c, co, u, uo = off["c"][q][:P], off["co"][q][:P], off["u"][q][:P], off["uo"][q][:P]
        two = None
        if have_ref:
            two = []
            for so in libs:
                refso._SO = so; refso._lib = None
                r = refso.fast_surf(n, kind, m[0], m[1], m[2], m[3], m[4], per, P)
                two.append((r[2][:P], r[0][:P]) if kind == 2 else (r[3][:P], r[1][:P]))
            refso._SO = libs[0]; refso._lib = None
        # the oracle under "equally valid" fp32 evaluations of the reference's own formulas (oracle/surfdisp_oracle.c, g_variant:
        # exponentials through exp2f, flattening factors from double-precision log / pow): does one of them return what the
        # HIP path returned?  Then the result is decided by the rounding of the reference's libm, not by its formulas
        var = []
        for v in (1, 2, 3):
            cport.lib().surfdisp_oracle_set_variant(v)
            cv, uv, sv = cport.forward_batch(m[None], per, kind)
            var.append((cv[0], uv[0]))
        cport.lib().surfdisp_oracle_set_variant(0)
        # ... and the statement-by-statement kernel (SURFDISP_STRICT: the reference's formulas in the reference's order, on the GPU,
        # with the GPU's exp / sin / cos): where IT leaves the oracle too, the last bit of the libm decides the result
        cs, us, ss = forward.forward_batch(m[None], per, kind, strict=True)
        cs, us = cs[0], us[0]
        k = 0
        with np.errstate(all="ignore"):
            if cat == 0:
                why = "zero pattern: UNEXPLAINED"
                if two and not np.array_equal(two[0][0] > 0, two[1][0] > 0): why = "zero pattern: the reference's builds differ too"
                elif any(np.array_equal(cv > 0, c > 0) for cv, uv in var): why = "zero pattern: as the oracle's with another rounding of exp / flattening"
                elif not np.array_equal(cs > 0, co > 0): why = "zero pattern: the statement-by-statement kernel (GPU libm) leaves the oracle too"
                else:
                    kd = int(np.nonzero((c > 0) != (co > 0))[0][0])      # first period with a root on one side only
                    kq = kd if max(c[kd], co[kd]) > 0 else kd
                    roots = [r for r in (c[kd], co[kd], c[kd - 1] if kd else 0.0, co[kd - 1] if kd else 0.0) if r > 0]
                    if any(near_scan_point(m, n, kind, per, P, kk, r) for r in roots for kk in ({kd, max(kd - 1, 0)})):
                        why = "zero pattern: a scan trial whose sign is within rounding (root within 3e-6 of it, or |value| < 2e-3 of both neighbours)"
            elif cat == 1:
                bad = np.abs(c / co - 1) > 1e-4
                k = int(np.nanargmax(np.abs(c / co - 1)))
                sp = abs(two[1][0][k] / two[0][0][k] - 1) if two and two[0][0][k] > 0 else 0.0
                why = "c: UNEXPLAINED"
                if sp > 1e-4: why = "c: the reference's builds disagree"
                elif any((np.abs(c[bad] / cv[bad] - 1) < 2e-5).all() for cv, uv in var if (cv[bad] > 0).all()):
                    why = "c: as the oracle's with another rounding of exp / flattening"
                elif (np.abs(cs[bad] / co[bad] - 1) > 1e-4).any(): why = "c: the statement-by-statement kernel (GPU libm) leaves the oracle too"
                else:
                    kd = int(np.nonzero(np.nan_to_num(np.abs(c / co - 1)) > 2e-5)[0][0])      # first period that differs
                    # (the period before counts too: a bracket one grid step further on there leaves the same root but another
                    # mmax behind, and the next period then rebuilds another number of layers for its T - calcul.f's carry-over)
                    if near_scan_point(m, n, kind, per, P, kd, co[kd]) or near_scan_point(m, n, kind, per, P, kd, c[kd]) or \
                       (kd > 0 and near_scan_point(m, n, kind, per, P, kd - 1, co[kd - 1])):
                        why = "c: a scan trial whose sign is within rounding (root within 3e-6 of it, or |value| < 2e-3 of both neighbours); next overtone 0.01 km/s on"
            else:
                e = np.nan_to_num(np.abs(u / uo - 1), nan=9.0)
                e = np.where(~np.isfinite(uo) & ~np.isfinite(u), 0.0, e)
                k = int(np.argmax(e))
                if not (np.isfinite(uo[k]) and np.isfinite(u[k])):
                    why = "U: overflow regime (NaN on one side)"
                else:
                    _, up = cport.group_at(m[None], per, kind, (co * np.float32(1 + DC))[None])
                    _, um = cport.group_at(m[None], per, kind, (co * np.float32(1 - DC))[None])
                    cond = max(abs(float(up[0][k]) - uo[k]), abs(float(um[0][k]) - uo[k])) / abs(uo[k])
                    sp = abs(two[1][1][k] / two[0][1][k] - 1) if two else 0.0
                    if not np.isfinite(cond) or cond > 1e-4: why = "U: ill conditioned (|dlnU/dlnc| x 2e-6 > 1e-4)"
                    elif not np.isfinite(sp) or sp > 2e-5: why = "U: the reference's builds disagree"
                    elif any(abs(u[k] / uv[k] - 1) < 5e-5 for cv, uv in var if uv[k] != 0): why = "U: as the oracle's with another rounding of exp / flattening"
                    elif abs(us[k] / uo[k] - 1) > 1e-4: why = "U: the statement-by-statement kernel (GPU libm) leaves the oracle too"
                    else: why = "U: UNEXPLAINED"
        counts[why] = counts.get(why, 0) + 1
        if "UNEXPLAINED" in why and len(unexpl) < 12:
            unexpl.append((q, kind, off["team"][q], n, float(per[k]), float(c[k]), float(co[k]), float(u[k]), float(uo[k])))
    print("soak offenders by cause (of the first", min(len(off["cat"]), int(os.environ.get("SOAK_CLASSIFY", "2500"))), "saved;",
          "reference builds " + ("available" if have_ref else "NOT available") + "):")
    for kq, v in sorted(counts.items(), key=lambda x: -x[1]): print(f"    {v:6d}  {kq}")
    for t in unexpl: print("    unexplained: saved #%d kind=%d team=%d L=%d T=%.4f  c %.6f (oracle %.6f)  U %.6f (oracle %.6f)" % t)
if off["cat"]:
    classify()
print(f"soak stacks with the reference's zero pattern but a phase velocity off by > 1e-4: {nbigc} ({nbigc / max(nstack, 1):.2e}); "
      f"with every phase velocity within 1e-4 but a group velocity off by > 1e-4: {nbigu} ({nbigu / max(nstack, 1):.2e}); "
      f"offending stacks saved: {len(off['cat'])}")
print(f"soak ({'default vs strict mode, GPU only' if STRICT else 'HIP vs CPU oracle'}): {ncase} cases, {nstack} stacks, zero-pattern mismatches {npat} stacks "
      f"({npat / max(nstack, 1):.2e}; {novf} of them where the secular function overflowed fp32: SURFDISP_NUMERIC "
      f"or the oracle's NEVILL failure), worst c {worst['c']:.2e}, worst U(99.9%) {worst['u']:.2e}")
bad_cases.sort(key=lambda t: (-t[10], -t[7]))
for b in bad_cases[:40]:
    print("  flagged: L=%d B=%d kind=%d noise=%.2f mono=%s P=%d team=%d  c %.1e  Umax %.1e  U99.9 %.1e  pattern %d" % b)
