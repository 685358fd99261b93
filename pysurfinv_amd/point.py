"""Drop-in for the reference's ``Point`` (``point.py:8-128``): one surface location with observed
Rayleigh phase velocities, inverted by Metropolis sampling - with every chain advanced in lock step on the
GPU (``MetropolisBatch``) instead of one process per chain.

    p = Point(setting, localInfo={}, periods=T, vels=c, uncers=sigma)
    p.misfit()                                   # (misfit, chiSqr, L) of the initial model, point.py:15-31
    p.MCinvMP(outdir, pid, runN=50000, chainL=1000, seed=42)        # -> {outdir}/{pid}.npz

The ``.npz`` has the reference's keys (``mcTrack`` rows ``[misfit, L, accepted, *params]`` in
``MCinv._brownians()`` order, ``setting``, ``obs``, ``invMeta``; ``point.py:82-85,120-123``), so
``PostPoint`` / ``Model3D.loadInvDir`` read it unchanged.  Differences, on purpose: ``nprocess`` is
ignored (all ``runN // chainL`` chains run at once on one device); random numbers come from a
``torch.Generator`` seeded with ``seed``, not from CPython's ``random`` (the reference's exact stream is
reproduced by ``brownian.PythonRandomProposer`` for single chains, see ``tests/test_mcmc.py``);
``localInfo`` is merged into ``Info`` as ``Model1D._loadLocalInfo`` does - the Cascadia-specific model
types, which rewrite layers from it (``models.py:528-575``), are out of scope.
"""
from __future__ import annotations

import copy

import numpy as np

from .layers_batch import Model1DBatch
from .mcmc import MetropolisBatch


class Point:
    def __init__(self, setting=None, localInfo={}, modelTypeCustom=None, layerClassCustom={},
                 periods=[], vels=[], uncers=[], device="cuda:0"):
        if modelTypeCustom is not None or layerClassCustom:
            raise NotImplementedError("custom model / layer classes are Python callbacks of the reference; "
                                      "Model1DBatch supports the built-in layer types")
        setting = copy.deepcopy(setting)
        setting.setdefault("Info", {}).update(localInfo)           # Model1D._loadLocalInfo, models.py:54-59
        self.setting = setting
        self.device = device
        self.initMod = Model1DBatch(setting, device=device)
        self.obs = {"T": periods, "c": vels, "uncer": uncers}      # Rayleigh wave, phase velocity only
        self.pid = "test"
        self._mc = None

    def _sampler(self, seed=None, **kw):
        return MetropolisBatch(self.initMod.spec, self.initMod.to_model, self.obs["T"], self.obs["c"],
                               self.obs["uncer"], device=self.device, seed=seed, **kw)

    def misfit(self, params=None):
        """(misfit, chiSqr, L) of one parameter vector (default: the initial model), point.py:15-31."""
        import torch
        mc = self._sampler()
        p = self.initMod.spec.v0 if params is None else params
        p = torch.as_tensor(np.asarray(p, float)[None, :], dtype=torch.float64, device=self.device)
        mis, chi, L = mc.misfit(p)
        return float(mis[0]), float(chi[0]), float(L[0])

    def MCinvMP(self, outdir="MCtest", pid=None, runN=50000, chainL=1000, nprocess=None, seed=42,
                priori=False, isgood=None, verbose=True, spec_depth=None, independent=False, fast_scan=False):
        """``runN // chainL`` chains of ``chainL`` steps each, the first one started at the initial model
        (point.py:91-125); writes ``{outdir}/{pid}.npz`` and returns the mcTrack array [runN, 3 + N].
        ``independent="auto"``: the period-parallel root search while the chains are too few to fill the chip (3 x faster
        lock steps for 100 chains; see ``MetropolisBatch``) - opt-in, the default is the reference's period walk.
        ``spec_depth``: None = the sampler's default (speculative lock steps on the device path: three Metropolis steps per
        batched solve for up to 292 chains, same chain distribution; ``MetropolisBatch.auto_spec_depth``), 1 = one step per solve."""
        if priori and outdir.split("_")[-1] != "priori":
            outdir = "_".join((outdir, "priori"))
        pid = self.pid if pid is None else pid
        mc = self._sampler(seed=seed, isgood=isgood, independent=independent, fast_scan=fast_scan)
        track = mc.run(max(int(runN) // int(chainL), 1), int(chainL), init_first=True, priori=priori,
                       spec_depth=spec_depth)
        arr = track.cpu().numpy().reshape(-1, track.shape[-1])
        MetropolisBatch.save_npz(outdir, pid, arr, self.setting, self.obs, chainL)
        return arr

    def MCinv(self, outdir="MCtest", pid=None, runN=50000, chainL=1000, init=True, seed=None, verbose=False,
              priori=False, isgood=None):
        """Same sampling as ``MCinvMP`` (the reference's serial loop, point.py:32-89, is ``runN // chainL``
        chains one after the other; here they run side by side)."""
        return self.MCinvMP(outdir, pid, runN, chainL, seed=seed, priori=priori, isgood=isgood, verbose=verbose)

    def copy(self):
        return copy.deepcopy(self)
