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
        ``spec_depth``: None = the sampler's default (speculative lock steps on the device path: four Metropolis steps per
        batched solve for up to 136 chains, three up to 292, same chain distribution; ``MetropolisBatch.auto_spec_depth``), 1 = one step per solve."""
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


class _ModelState:
    """What the post-processing keeps of a model: its random-walk parameters (``_brownians()`` order), misfit, likelihood."""

    def __init__(self, params, misfit=None, L=None):
        self.params = np.asarray(params, float)
        self.misfit, self.L = misfit, L

    def _brownians(self):
        return list(self.params)


class PostPoint(Point):
    """Drop-in for the compute part of the reference's ``PostPoint`` (point.py:134-175, 307-335): reads one point's
    ``{pid}.npz`` (keys ``mcTrack, setting, obs, invMeta`` - the reference's files and this package's alike) and derives

    * ``MCparas`` with rejected rows replaced by the last accepted row before them (``trueMarkovChain``, :154-159);
    * ``minMod`` = the row of the smallest misfit (:161-165), ``thres = max(2 min, min + 0.5)`` (:307-309),
      ``accFinal = misfits < thres`` (:167-168);
    * ``avgMod`` = mean parameters of the final rows (:170-171) with its misfit and likelihood from ONE forward solve on
      the device (:173);
    * ``_loadValues(indVars)`` = the final rows' parameters (:317-335, the parameter branch), ``_model_generator``.

    ``minMod`` / ``avgMod`` carry ``params`` (``_brownians()`` order), ``misfit``, ``L``; ``initMod`` is the point's
    ``Model1DBatch`` (``seis_prop_layers(params)`` / ``seis_prop_grids(params)`` give any of them as a profile).  The plotting
    methods of the reference are out of scope.  ``device=None``: no forward solve (``avgMod.misfit`` stays None)."""

    def __init__(self, npzMC=None, npzPriori=None, modelTypeCustom=None, layerClassCustom={}, trueMarkovChain=True,
                 device="cuda:0", _forward=None):
        self.MCparas_pri = None
        self._forward = _forward
        if npzMC is not None:
            tmp = np.load(npzMC, allow_pickle=True)
            self.MC, setting, obs = np.array(tmp["mcTrack"], float), tmp["setting"][()], tmp["obs"][()]
            self.invMeta = tmp["invMeta"][()]
            Point.__init__(self, setting, modelTypeCustom=modelTypeCustom, layerClassCustom=layerClassCustom,
                           periods=obs["T"], vels=obs["c"], uncers=obs["uncer"], device=device if device is not None else "cpu")
            self.pid = self.invMeta.get("pid", self.pid) if isinstance(self.invMeta, dict) else self.pid
            self.N = self.MC.shape[0]
            self.misfits, self.Ls, self.accepts = self.MC[:, 0], self.MC[:, 1], self.MC[:, 2]
            self.MCparas = self.MC[:, 3:]
            if trueMarkovChain and self.N:
                idx = np.where(self.accepts.astype(bool), np.arange(self.N), -1)
                last = np.maximum.accumulate(idx)
                if last[0] < 0:                                       # the reference would fail here too (iAcc undefined)
                    raise ValueError("mcTrack does not start with an accepted row")
                self.MCparas = self.MCparas[last]
                self.MC = np.concatenate([self.MC[:, :3], self.MCparas], axis=1)
            ind_min = int(np.nanargmin(self.misfits))
            self.minMod = _ModelState(self.MCparas[ind_min], float(self.misfits[ind_min]), float(self.Ls[ind_min]))
            self.thres = self._thres(self.minMod.misfit)
            self.accFinal = self.misfits < self.thres
            self.avgMod = _ModelState(self.MCparas[self.accFinal].mean(axis=0))
            if device is not None or _forward is not None:
                self.avgMod.misfit, _, self.avgMod.L = self.misfit(self.avgMod.params)
        if npzPriori is not None:
            self.MCparas_pri = np.array(np.load(npzPriori, allow_pickle=True)["mcTrack"], float)[:, 3:]

    def _sampler(self, seed=None, **kw):
        if self._forward is not None:
            kw.setdefault("forward", self._forward)
        return Point._sampler(self, seed=seed, **kw)

    @staticmethod
    def _thres(minMisfit):
        return max(minMisfit * 2, minMisfit + 0.5)

    def _model_generator(self, indSteps=None, priori=False):
        """Parameter vectors of the final rows (or of ``indSteps``; ``priori``: every row of the prior run)."""
        paras = self.MCparas if not priori else self.MCparas_pri
        if indSteps is None:
            indSteps = np.where(self.accFinal)[0] if not priori else range(len(self.misfits))
        for ind in indSteps:
            yield _ModelState(paras[ind])

    def _loadValues(self, indVars="all", zdeps=None, indSteps=None, priori=False):
        """[n_vars, n_final] parameter values of the final rows; with ``zdeps``: [len(zdeps), n_rows] Vs at those depths of
        the models of the final rows (or of ``indSteps``) - ``Model1D.value`` for all of them in one batched call where the
        reference maps a process pool over them (point.py:317-335)."""
        if zdeps is not None:
            paras = np.array([m.params for m in self._model_generator(indSteps, priori=priori)])
            import torch
            return self.initMod.value(torch.as_tensor(paras, dtype=torch.float64, device=self.initMod.device), zdeps).T
        indVars = range(self.MCparas.shape[1]) if isinstance(indVars, str) and indVars == "all" else indVars
        paras = self.MCparas[self.accFinal] if not priori else self.MCparas_pri[self.accFinal]
        return np.array([mc[list(indVars)] for mc in paras]).T
