"""Prior predicates (MCinv.perturb / reset redraw until isgood(model), models.py:192-219): the generic tests the
reference's model classes build `isgood` from (models.py:294-320) as `PriorRules` - torch evaluation on CPU against a numpy
restatement on reference-generated grid points (tests/golden/ref_grids.npz), the device kernel against the torch evaluation,
and the fused lock step with the masked redraw rounds (GPU)."""
import os

import numpy as np
import pytest
import torch

from pysurfinv_amd.layers_batch import Model1DBatch
from pysurfinv_amd.mcmc import MetropolisBatch, PriorRules
from pysurfinv_amd.settings import MCMC_PERIODS, MCMC_SETTING

from test_layers_batch import CONT, GG


def _numpy_rules(vs, names, monotone=("sediment", "crust"), jumps=True, vs_max=None):
    """models.py:302-320, statement by statement, on one model's grid points"""
    names = np.asarray(names)
    if jumps:
        for i in np.where(names[1:] != names[:-1])[0]:
            if vs[i + 1] < vs[i]:
                return False
    if vs_max is not None and np.any(vs > vs_max):
        return False
    for g in monotone:
        a = vs[names == g]
        if not np.all(np.diff(a) >= np.finfo(float).eps):
            return False
    return True


def test_rules_in_torch_equal_the_reference_statements_on_reference_grids():
    m = Model1DBatch(CONT)
    params = torch.from_numpy(GG["cont/params"])
    ref = GG["cont/grids"]
    groups = list(GG["groups"])
    rules = PriorRules(m)
    got = rules(params).numpy()
    # the fixture's grids hold Info.refLayer's reference mantle too: the rules see the model's own layers (refLayer = False)
    (z, vs, *_), grp, ngrid = m.seis_prop_grids(params, ref_layer=False)
    n = int(ngrid[0])
    want = []
    for i in range(params.shape[0]):
        assert np.abs(ref[i, 1, :n] - vs[i, :n].numpy()).max() < 1e-9          # (same points as the reference's, minus its tail)
        names = [groups[int(q)] for q in ref[i, 6, :n]]
        want.append(_numpy_rules(ref[i, 1, :n], names))
    assert np.array_equal(got, np.array(want))
    # a cap below the mantle's velocities rejects everything; none accepts what the other two rules accept
    assert not PriorRules(m, vs_max=3.0)(params).any()
    assert np.array_equal(PriorRules(m, vs_max=9.0)(params).numpy(), got)
    # more draws: the rules do discriminate
    from pysurfinv_amd.brownian import TorchProposer
    p = TorchProposer(m.spec, "cpu", seed=5).reset(400)
    g = rules(p).numpy()
    assert 0.02 < g.mean() < 0.98
    (z, vs, *_), grp, ngrid = m.seis_prop_grids(p, ref_layer=False)
    names = [Model1DBatch.GROUP_NAMES[int(q)] for q in grp[0, :int(ngrid[0])]]
    assert np.array_equal(g, np.array([_numpy_rules(vs[i, :int(ngrid[0])].numpy(), names) for i in range(400)]))


@pytest.mark.gpu
def test_prior_kernel_equals_the_torch_rules():
    import ctypes
    from pysurfinv_amd import _lib
    from pysurfinv_amd.brownian import TorchProposer
    dev = torch.device("cuda:0")
    for setting, kw in ((CONT, {}), (MCMC_SETTING, dict(vs_max=4.9)), (MCMC_SETTING, dict(monotone_groups=("crust",), positive_jumps=False))):
        m = Model1DBatch(setting, device=dev)
        rules = PriorRules(m, **kw)
        flags = rules.device_flags()
        assert flags is not None
        p = TorchProposer(m.spec, dev, seed=11).reset(5000).contiguous()
        want = rules(p)
        idesc, fdesc, L = m.native_descriptor()
        tags = torch.zeros(p.shape[0], dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().surfdisp_prior_device(ctypes.c_void_p(stream), p.shape[0], p.shape[1], int(L), ctypes.c_void_p(p.data_ptr()),
                                                    ctypes.c_void_p(idesc.data_ptr()), ctypes.c_void_p(fdesc.data_ptr()),
                                                    ctypes.c_void_p(flags.data_ptr()), ctypes.c_double(-1.0 if rules.vs_max is None else rules.vs_max),
                                                    -1, 1, ctypes.c_void_p(tags.data_ptr())))
        torch.cuda.synchronize()
        assert torch.equal(tags == 0, want), (kw, int((tags == 0).sum()), int(want.sum()))
        assert 0.01 < float(want.float().mean()) < 0.99


@pytest.mark.gpu
def test_fused_lock_step_with_prior_rules():
    """MetropolisBatch(isgood=PriorRules): the fused path stays on (fused_available), every recorded row satisfies the rules,
    and the acceptance statistics are those of the torch loop with the same predicate (the reference's redraw-until-good)."""
    from pysurfinv_amd.settings import synthetic_observations
    dev = torch.device("cuda:0")
    mb, c_obs, unc = synthetic_observations(1, dev, seed=100)
    rules = PriorRules(mb, vs_max=4.9)
    C, chainL = 512, 40
    out = {}
    for name, fused in (("fused", True), ("torch", False)):
        mc = MetropolisBatch(mb.spec, mb.to_model, MCMC_PERIODS, c_obs[0], unc[0], device=dev, seed=3, isgood=rules)
        assert mc.fused_available()
        tr = mc.run(C, chainL, init_first=False, fused=fused)
        torch.cuda.synchronize()
        rows = tr[:, :, 3:].reshape(-1, mb.spec.n)
        good = rules(rows)
        assert bool(good.all()), (name, int((~good).sum()))
        out[name] = (float(tr[:, 1:, 2].mean()), float(tr[:, -1, 0].median()))
    # same sampler statistically: acceptance rate and the misfit reached after 40 steps
    assert abs(out["fused"][0] - out["torch"][0]) < 0.03, out
    assert abs(out["fused"][1] / out["torch"][1] - 1) < 0.15, out
    # ... and the chains do move (a rejected chain would repeat its row)
    mc = MetropolisBatch(mb.spec, mb.to_model, MCMC_PERIODS, c_obs[0], unc[0], device=dev, seed=3, isgood=rules)
    tr = mc.run(4096, 6, init_first=False)                     # chain groups (two from 4 096 chains on)
    torch.cuda.synchronize()
    assert bool(rules(tr[:, :, 3:].reshape(-1, mb.spec.n)).all())
    assert float((tr[:, 1:, 3:] != tr[:, :-1, 3:]).any(dim=2).float().mean()) > 0.9
