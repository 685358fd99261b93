#!/usr/bin/env python3
"""Golden vectors for the compute part of the reference's ``PostPoint`` (point.py:134-175,307-335) from the imported
reference Python (development container only; see _refimport.py): the 240-row Metropolis trace of ref_driver.npz is written as
the ``.npz`` the reference's ``Point.MCinv`` writes, the reference's ``PostPoint`` loads it, and what it derives is recorded -
the parameters after the true-Markov-chain substitution, minimum-misfit model, threshold, final acceptance mask, average
model with its misfit and likelihood, ``_loadValues()``.

    python tests/golden/make_golden_post.py        ->  tests/golden/ref_post.npz  (+ the input file post_trace.npz)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

_refimport.install()
from pySurfInv.point import PostPoint               # noqa: E402
from settings import CONT                            # noqa: E402


ZDEPS = np.linspace(0.5, 180.0, 37)


def main():
    G = np.load(os.path.join(HERE, "ref_driver.npz"))
    obs = {"T": list(G["trace/periods"]), "c": list(G["trace/c_obs"]), "uncer": list(G["trace/uncer"])}
    f = os.path.join(HERE, "post_trace.npz")
    np.savez_compressed(f, mcTrack=G["trace/mcTrack"], setting=dict(CONT), obs=obs, invMeta={"pid": "trace", "chainL": 80})
    for tmc in (True, False):
        p = PostPoint(f, trueMarkovChain=tmc)
        out = dict(MCparas=p.MCparas, min_params=np.array(p.minMod._brownians()), min_misfit=p.minMod.misfit, min_L=p.minMod.L,
                   thres=p.thres, accFinal=p.accFinal, avg_params=np.array(p.avgMod._brownians()), avg_misfit=p.avgMod.misfit,
                   avg_L=p.avgMod.L, values=p._loadValues(), values_sub=p._loadValues(indVars=[0, 3, 7]),
                   # _loadValues(zdeps=...) without its process pool: the expression the reference keeps beside it (point.py:327)
                   values_z=np.array([mod.value(ZDEPS) for mod in p._model_generator()]).T)
        print("trueMarkovChain", tmc, "min misfit", p.minMod.misfit, "thres", p.thres, "final", int(p.accFinal.sum()), "avg misfit", p.avgMod.misfit)
        if tmc:
            res = {f"tmc/{k}": np.asarray(v) for k, v in out.items()}
        else:
            res.update({f"raw/{k}": np.asarray(v) for k, v in out.items()})
    res["zdeps"] = ZDEPS
    np.savez_compressed(os.path.join(HERE, "ref_post.npz"), **res)


if __name__ == "__main__":
    main()
