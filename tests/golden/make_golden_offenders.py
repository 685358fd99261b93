#!/usr/bin/env python3
"""Stacks on which the round-4 differential soak (scripts/soak.py, HIP path against the CPU oracle, r03 library) returned the
reference's zero pattern but ANOTHER ROOT (a phase velocity more than 1e-4 off) -> tests/golden/ref_offenders.npz, with what
the unmodified reference itself returns for them (oracle/_ref/libfast_surf_ref.so, fresh-process state).

Two causes were found from these stacks and fixed:
  * general family (water layer over soft sediments, first period of a few seconds, |Delta| ~ 1e19 .. 1e20): both three-point
    estimates of the refinement evaluated to exactly 0 - v_rcp_f32 flushes for denominators beyond 2^126 - "agreed", and the
    bracket's low end came back as the root (teams of fewer than 16 lanes);
  * soft-sediment family, Love, periods of 0.3 .. 3 s: several overtones inside one 0.01 km/s bracket, invisible to the
    subdivision of a small team - the reference's NEVILL lands on one of them by its own evaluation sequence.
Each stack carries `defined`: the reference's second build (FMA contraction) and the oracle under two other roundings of its
own formulas (exp through exp2f; flattening factors from double-precision log / pow) all return the same roots to 2e-5 - only
those stacks can be held to the bar by an arithmetic that is not the reference's own, bit for bit; the others document what a
rounding-decided root looks like.  Build container only (needs /root/reference through oracle/_ref and the soak's offender files):

    python tests/golden/make_golden_offenders.py tmp_probe/soak_offenders_general_oracle.npz tmp_probe/soak_offenders_sediment_oracle.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cport, refso                            # noqa: E402

CAPS = {"general": 70, "sediment": 90}


def main(files):
    libs = [refso._SO, refso._SO.replace("_ref.so", "_ref_fma.so")]
    rows = []
    for fn in files:
        fam = "sediment" if "sediment" in fn else "general"
        f = np.load(fn)
        idx = np.nonzero(f["cat"] == 1)[0]
        idx = idx[:: max(1, len(idx) // CAPS[fam])][: CAPS[fam]]
        for q in idx:
            n, P, kind = int(f["nlay"][q]), int(f["P"][q]), int(f["kind"][q])
            m = np.ascontiguousarray(f["model"][q][:, :n]); per = f["per"][q][:P].copy()
            two = []
            for so in libs:
                refso._SO = so; refso._lib = None
                r = refso.fast_surf(n, kind, m[0], m[1], m[2], m[3], m[4], per, P)
                two.append((r[2][:P].copy(), r[0][:P].copy()) if kind == 2 else (r[3][:P].copy(), r[1][:P].copy()))
            refso._SO = libs[0]; refso._lib = None
            cr, ur = two[0]
            alts = [two[1][0]]
            for v in (1, 2, 3):
                cport.lib().surfdisp_oracle_set_variant(v)
                alts.append(cport.forward_batch(m[None], per, kind)[0][0])
            cport.lib().surfdisp_oracle_set_variant(0)
            co = cport.forward_batch(m[None], per, kind)[0][0]
            assert np.array_equal(co, cr), "the oracle is pinned bit for bit to the reference"
            with np.errstate(all="ignore"):
                defined = all(np.array_equal(a > 0, cr > 0) and (np.abs(a[cr > 0] / cr[cr > 0] - 1) < 2e-5).all() for a in alts)
            rows.append(dict(fam=fam, model=f["model"][q], nlay=n, per=f["per"][q], P=P, kind=kind, team=int(f["team"][q]),
                             c=np.pad(cr, (0, 40 - P)), u=np.pad(ur, (0, 40 - P)), defined=defined, c_r03=f["c"][q]))
    out = {k: np.array([r[k] for r in rows]) for k in rows[0]}
    out["flang"] = np.array(open(os.path.join(os.path.dirname(libs[0]), "BUILD_INFO.txt")).read())
    np.savez_compressed(os.path.join(HERE, "ref_offenders.npz"), **out)
    d = out["defined"]
    for fam in ("general", "sediment"):
        s = out["fam"] == fam
        print(fam, "stacks", int(s.sum()), "defined", int((d & s).sum()), "kinds", np.bincount(out["kind"][s], minlength=3)[1:])


if __name__ == "__main__":
    main(sys.argv[1:])
