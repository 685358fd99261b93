"""Batched, device-resident Metropolis sampler: the GPU counterpart of ``Point.misfit`` /
``Point.MCinv`` / ``Point.MCinvMP`` (``/root/reference/point.py:15-125``).

The reference advances one chain per process, one forward solve per step.  Here C chains (of one or
many grid points) advance in lock step: every step is ONE batched forward solve of C stacks through
``libsurfdisp_hip`` (root-search kernel only - the misfit uses Rayleigh phase velocity,
``point.py:11,18``), with proposals, misfit and the accept rule evaluated by torch ops on the device.

Semantics kept from the reference:
* proposal: every random-walk scalar moves by a bounded Gaussian step (``brownian.py:20-27``), the
  whole proposal is redrawn while ``isgood`` rejects it (<= 1000 times, then uniform ``reset``,
  ``models.py:192-219``);
* misfit: ``chi2 = sum(((cO-cP)/uncer)**2)``, ``misfit = sqrt(chi2/N)``, ``chi2 := sqrt(50*chi2)``
  when ``chi2 >= 50``, ``L = exp(-chi2/2)``; a failed forward solve gives ``(88888, 88888, 0)``
  (``point.py:20-31``);
* accept: ``chi1 < chi0`` or ``random() > 1 - exp(-(chi1-chi0)/2)`` (``point.py:34-37``);
* a chain starts from the initial model (first chain when ``init``) or from a uniform prior draw
  (``point.py:47-57``); ``mcTrack`` rows are ``[misfit, L, accepted, *params]`` of the PROPOSED
  model (``models.py:254-256``, ``point.py:58,73-76``); output ``.npz`` keys ``mcTrack, setting,
  obs, invMeta`` (``point.py:82-85``) so ``PostPoint`` / ``Model3D.loadInvDir`` can read it.
RNG: CPython's Mersenne Twister cannot be reproduced on the device; the production path uses a
torch Philox generator (statistical parity), and ``PythonRandomProposer`` replays the reference's
exact stream for one chain (trace parity, ``tests/test_mcmc.py``).
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib
from .brownian import ParamSpec, TorchProposer

FAIL = 88888.0                                                  # point.py:21


class PriorRules:
    """The GENERIC prior predicates the reference's model classes build ``isgood`` from (models.py:294-320), in a form the
    device can evaluate (csrc/surfdisp_layers.hip::surfdisp_prior_kernel) - pass it as ``MetropolisBatch(isgood=...)`` and
    the lock step stays on the device (fused kernels, chain groups, HIP graphs):

    * ``monotone_groups``: Vs must increase with depth over the grid points of these layer groups ("sediment", "crust", ...:
      ``monoIncrease``, models.py:8-9, 315-320);
    * ``positive_jumps``: Vs must not drop across a boundary between two groups (models.py:302-307);
    * ``vs_max``: every grid point's Vs at most this (models.py:309-313; None: no cap).

    The reference redraws the WHOLE proposal until the predicate holds (``MCinv.perturb``: 1000 tries, then ``reset``'s uniform
    draws, models.py:192-219).  On the device the same loop runs as MASKED redraw rounds without a host synchronisation:
    ``rounds`` Gaussian rounds (only the chains whose proposal broke a rule draw again), then ``reset_rounds`` rounds of uniform
    prior draws, and a chain that is still without a good proposal proposes its own state (a wasted step: with a prior that
    accepts half of the Gaussian draws, 1 chain in 60 reaches the uniform rounds and fewer than 1 in 1e3 the wasted step).  Called with a parameter tensor it evaluates
    the same rules in torch on ``Model1DBatch.seis_prop_grids`` (the torch lock step; the tests compare the two)."""

    def __init__(self, model, monotone_groups=("sediment", "crust"), positive_jumps=True, vs_max=None, rounds=5, reset_rounds=2):
        self.model = model
        self.monotone_groups = tuple(monotone_groups)
        self.positive_jumps = bool(positive_jumps)
        self.vs_max = None if vs_max is None else float(vs_max)
        self.rounds, self.reset_rounds = int(rounds), int(reset_rounds)
        self._flags = None

    def __call__(self, params, rows=None):
        return self.model.prior_good(params, self, rows=rows)

    def device_flags(self):
        """int32 [L] flags for the prior kernel, or None (no native descriptor / thermal layer: the torch loop serves)."""
        if self._flags is None:
            self._flags = self.model.prior_flags(self)
        return self._flags


class ChainGroups:
    """The chains of a ``MetropolisBatch`` as contiguous groups, each advancing on its own stream (``chain_groups``).
    ``fork()`` once (the group streams wait for the current stream), ``step()`` per lock step, ``join()`` once (the current
    stream waits for every group): between the two nothing synchronises the groups with each other."""

    def __init__(self, mc, C, G):
        import copy
        torch = mc.torch
        self.mc, self.C, self.G = mc, int(C), int(G)
        self.bounds = [self.C * g // self.G for g in range(self.G + 1)]
        self.streams = [torch.cuda.Stream(device=mc.device) for _ in range(self.G)]
        self.children = []
        for g in range(self.G):
            lo, hi = self.bounds[g], self.bounds[g + 1]
            ch = copy.copy(mc)                                  # shares spec, proposer (bounds, steps, seed), periods, to_model
            ch._plan, ch._fz, ch.event_ring, ch._ev_i, ch.n_forward, ch._groups, ch._plans = None, None, None, 0, 0, {}, {}
            if mc.c_obs.ndim == 2:
                if mc.c_obs.shape[0] != self.C:
                    raise ValueError(f"{self.C} chains against {mc.c_obs.shape[0]} rows of observations")
                ch.c_obs, ch.uncer, ch.mask = mc.c_obs[lo:hi], mc.uncer[lo:hi], mc.mask[lo:hi]
            if mc.local_rows is not None:
                ch.local_rows = mc.local_rows[lo:hi]
            ch._chain0 = mc._chain0 + lo
            ch._pipelined = True                                # the library sizes its teams for both groups' stacks
            self.children.append(ch)

    def _refresh(self):
        """The children's slices of the parent's observations / local rows and its run-time switches, taken afresh at every
        fork: a parent whose observations or ``local_rows`` were replaced, or whose ``independent`` / ``fast_scan`` changed after
        a first grouped run, must not advance on the first run's copies."""
        mc = self.mc
        for g, ch in enumerate(self.children):
            lo, hi = self.bounds[g], self.bounds[g + 1]
            if mc.c_obs.ndim == 2:
                if mc.c_obs.shape[0] != self.C:
                    raise ValueError(f"{self.C} chains against {mc.c_obs.shape[0]} rows of observations")
                ch.c_obs, ch.uncer, ch.mask = mc.c_obs[lo:hi], mc.uncer[lo:hi], mc.mask[lo:hi]
            else:
                ch.c_obs, ch.uncer, ch.mask = mc.c_obs, mc.uncer, mc.mask
            ch.local_rows = None if mc.local_rows is None else mc.local_rows[lo:hi]
            ch.independent, ch.fast_scan, ch.isgood = mc.independent, mc.fast_scan, mc.isgood
            ch._chain0 = mc._chain0 + lo
            if ch._fz is not None and (ch._fz["c_obs"].data_ptr() != ch.c_obs.contiguous().data_ptr()):
                ch._fz = None                                   # (the fused buffers hold contiguous copies of the observations)

    def fork(self):
        self._refresh()
        cur = self.mc.torch.cuda.current_stream(self.mc.device)
        for s in self.streams:
            s.wait_stream(cur)
        self.children[0].event_ring, self.children[0]._ev_i = self.mc.event_ring, self.mc._ev_i    # bench.py's measurement hook

    def step(self, p, row=None, row_stride=0, row_offset=0, first=False, counter=None):
        """One Metropolis step of every chain: state ``p`` [C, N] in place, rows as ``MetropolisBatch.fused_step``."""
        torch = self.mc.torch
        if counter is None:
            self.mc._counter += 1
            counter = self.mc._counter
        for g, ch in enumerate(self.children):
            lo, hi = self.bounds[g], self.bounds[g + 1]
            with torch.cuda.stream(self.streams[g]):
                ch.fused_step(p[lo:hi], row=row, row_stride=row_stride, row_offset=row_offset + lo * row_stride,
                              first=first, counter=counter)

    def join(self):
        cur = self.mc.torch.cuda.current_stream(self.mc.device)
        for s in self.streams:
            cur.wait_stream(s)
        for ch in self.children:
            self.mc.n_forward += ch.n_forward
            ch.n_forward = 0
        self.mc._ev_i = self.children[0]._ev_i
        self.children[0].event_ring = None


class MetropolisBatch:
    """C chains in lock step.

    to_model(params[C, N] float64 tensor) -> (model[C, 5, L] float32 tensor, nlay[C] int32 or None)
    isgood(params) -> bool[C] tensor (None: always good, ``MCinv.isgood`` default, models.py:220-224)
    c_obs, uncer: [P] or [C, P]; entries with uncer <= 0 or NaN c_obs are masked out (the
    reference uses a masked array, point.py:23-26).
    """

    AUTO_INDEP_CHAINS = 3072        # 64-lane teams of fewer chains leave the chip's 196 608 lanes partly empty

    def __init__(self, spec: ParamSpec, to_model, periods, c_obs, uncer, device="cuda:0",
                 isgood=None, proposer=None, seed=None, forward=None, independent=False, fast_scan=False,
                 local_rows=None):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.spec = spec
        self.to_model = to_model
        self.isgood = isgood
        self.proposer = proposer if proposer is not None else TorchProposer(spec, self.device, seed)
        self.periods = torch.as_tensor(np.asarray(periods, np.float32), device=self.device)
        co = torch.as_tensor(np.asarray(c_obs, np.float64), device=self.device)
        un = torch.as_tensor(np.asarray(uncer, np.float64), device=self.device)
        self.mask = torch.isfinite(co) & torch.isfinite(un) & (un > 0)
        self.c_obs = torch.where(self.mask, co, torch.zeros_like(co))
        self.uncer = torch.where(self.mask, un, torch.ones_like(un))
        self._plan = None
        self._forward = forward                                 # test hook: callable(model, nlay) -> (c, status)
        # independent=True: period-parallel root search (SURFDISP_INDEPENDENT) - lower latency for few
        # chains; only for smooth parameterisations (no low-velocity roughness), see include/surfdisp.h.
        # independent="auto": that decomposition whenever the lock step is too small to fill the chip with whole-stack
        # teams (fewer than AUTO_INDEP_CHAINS chains: 100 chains x 96 layers take 0.92 ms per step in the faithful period
        # walk - a 19-period dependent chain on 100 of 4 096 wavefront slots - against 0.3 ms), the faithful walk
        # otherwise.  NOT the default: the two agree to 4e-6 with identical zero patterns on 40 000 prior draws of the
        # continental and the thermal model (scripts/indep_vs_faithful.py, profiles/r03a), but on rough stacks the
        # reference's answer depends on its period list (start rule 0.9 c(k-1), stale deep layers: up to 2.4e-3,
        # tests/test_gpu_parity.py) and a default must not change results with the number of chains.
        if independent not in (True, False, "auto"):
            raise ValueError("independent must be True, False or 'auto'")
        self.independent = independent
        # fast_scan=True: opt into the heuristic coarse-to-fine scan (SURFDISP_FASTSCAN); the default walks every
        # grid point of the reference's scan, so root selection and failures are the reference's on every input
        self.fast_scan = bool(fast_scan)
        self.n_forward = 0
        # local_rows [C]: row of the model's per-point local-information table (Model1DBatch.set_local_info) each chain
        # belongs to; to_model is then called as to_model(params, rows)
        self.local_rows = None if local_rows is None else torch.as_tensor(np.asarray(local_rows), dtype=torch.int64,
                                                                          device=self.device)
        # measurement hook (bench.py): a forward.EventRing whose next slot brackets the solver's kernels of each call
        self.event_ring = None
        self._ev_i = 0
        self._counter = 0                                       # calls of the fused kernels so far: the Philox counter of the next one
        self._pipelined = False                                 # a chain group: another group's solve is in flight beside this one
        self._chain0 = 0                                        # index of this object's chain 0 in the whole sampler (chain groups)
        self._groups = {}

    # ------------------------------------------------------------------ forward + misfit
    def forward_c(self, params, rows=None):
        """Rayleigh phase velocities c[C, P] and status[C] for the stacks of ``params`` (``rows``: the chain each row
        of ``params`` belongs to, when it is not simply row i = chain i)."""
        torch = self.torch
        if self.local_rows is not None:
            if rows is None and params.shape[0] != self.local_rows.shape[0]:
                raise ValueError(f"{params.shape[0]} models against {self.local_rows.shape[0]} chains with local information: pass rows=")
            model, nlay = self.to_model(params, self.local_rows if rows is None else self.local_rows[rows])
        else:
            model, nlay = self.to_model(params)
        if self._forward is not None:
            self.n_forward += model.shape[0]
            return self._forward(model, nlay)
        c, st = self._solve_model(model, nlay)
        return c.to(torch.float64), st

    def misfit(self, params, rows=None, return_c=False):
        """(misfit, chiSqr, L) per row of ``params`` - point.py:15-31.  With per-chain observations (``c_obs``
        [C, P]) row i of ``params`` is compared with observation row i, or with row ``rows[i]`` when ``rows`` (an
        index tensor) is given - the speculative sampler evaluates several proposals per chain, the grid driver
        one average model per point."""
        torch = self.torch
        cP, st = self.forward_c(params, rows)
        failed = (st != 0) | (cP < 0.01).any(dim=1)            # models.py:29-33
        c_obs, uncer, mask = self.c_obs, self.uncer, self.mask
        if c_obs.ndim == 2:
            if rows is not None:
                c_obs, uncer, mask = c_obs[rows], uncer[rows], mask[rows]
            elif c_obs.shape[0] != cP.shape[0]:
                raise ValueError(f"{cP.shape[0]} models against {c_obs.shape[0]} rows of observations: pass rows=")
        r = torch.where(mask, (c_obs - cP) / uncer, torch.zeros_like(cP))
        chi = (r * r).sum(dim=1)
        N = mask.sum(dim=-1).to(torch.float64)
        mis = torch.sqrt(chi / N)
        chi = torch.where(chi < 50, chi, torch.sqrt(chi * 50.0))
        L = torch.exp(-0.5 * chi)
        big = torch.full_like(chi, FAIL)
        out = (torch.where(failed, big, mis), torch.where(failed, big, chi),
               torch.where(failed, torch.zeros_like(L), L))
        return out + (cP,) if return_c else out

    # ------------------------------------------------------------------ fused device path (csrc/surfdisp_mcmc.hip)
    def fused_available(self):
        """The lock step can run as propose kernel -> stacks -> solver -> accept kernel (no torch glue, no host
        synchronisation): device proposer, the HIP forward path, and no ``isgood`` callback - or a ``PriorRules`` one, whose
        redraw loop runs on the device too (``_propose_with_rules``)."""
        dev_prior = self.isgood is None or (isinstance(self.isgood, PriorRules) and self.device.type == "cuda"
                                            and self.local_rows is None and self.isgood.device_flags() is not None)
        return (dev_prior and isinstance(self.proposer, TorchProposer) and self.device.type == "cuda"
                and self._forward is None)

    def _fused_buffers(self, C):
        torch = self.torch
        st = getattr(self, "_fz", None)
        if st is None or st["C"] != C:
            N = self.spec.n
            st = dict(C=C, p1=torch.empty((C, N), dtype=torch.float64, device=self.device),
                      chi=torch.zeros(C, dtype=torch.float64, device=self.device),
                      mask8=self.mask.to(torch.uint8).contiguous(), c_obs=self.c_obs.contiguous(), uncer=self.uncer.contiguous())
            self._fz = st
        return st

    def _solve_raw(self, params, rows=None):
        """fp32 c[C, P] and status[C] of the solver's own output tensors (no copies) for the stacks of ``params``."""
        if self.local_rows is not None:
            model, nlay = self.to_model(params, self.local_rows if rows is None else self.local_rows[rows])
        else:
            model, nlay = self.to_model(params)
        return self._solve_model(model, nlay)

    def _solve_model(self, model, nlay):
        self.n_forward += model.shape[0]
        from .forward import BatchPlan
        C, _, L = model.shape
        if self._plan is None or (self._plan.B, self._plan.L) != (C, L):
            # (the speculative lock step alternates between C and C * (2^d - 1) stacks: keep a plan per size)
            plans = self.__dict__.setdefault("_plans", {})
            if (C, L) not in plans:
                if len(plans) >= 4:
                    plans.clear()
                plans[(C, L)] = BatchPlan(C, L, self.periods.numel(), device=self.device)
            self._plan = plans[(C, L)]
        ev = None
        if self.event_ring is not None:
            ev = self.event_ring.slot(self._ev_i)
            self._ev_i += 1
        indep = (C < self.AUTO_INDEP_CHAINS) if self.independent == "auto" else bool(self.independent)
        c, _, st = self._plan.run(model.contiguous(), self.periods, kind=_lib.KIND_RAYLEIGH | _lib.PHASE_ONLY,
                                  nlay=nlay, independent=indep, fast_scan=self.fast_scan, events=ev,
                                  pipelined=self._pipelined)
        return c, st

    def fused_step(self, p, row=None, row_stride=0, first=False, counter=None, row_offset=0):
        """One Metropolis step of every chain, in place on the state ``p`` [C, N] (float64, contiguous): proposal
        (``first``: none - the states themselves are evaluated and accepted, a chain's first row), stacks, forward solve,
        misfit / accept / update.  ``row``: a float64 tensor whose element ``row_offset`` is where chain 0's mcTrack row
        goes, chain c's ``row_stride`` doubles further.  ``counter``: the Philox counter of this step (default: one more
        than the last call's).  Everything stream-ordered on the current stream."""
        import ctypes
        torch = self.torch
        C, N = p.shape
        st = self._fused_buffers(C)
        L = _lib.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
        pr = self.proposer
        if counter is None:
            self._counter += 1
            counter = self._counter
        rowp = ctypes.c_void_p(row.data_ptr() + 8 * int(row_offset)) if row is not None else ctypes.c_void_p(0)
        with torch.cuda.device(self.device):
            if first:
                p1 = p
            else:
                p1 = st["p1"]
                _lib.check(L.surfdisp_mcmc_propose_device(stream, C, N, ptr(p), ptr(pr.vmin), ptr(pr.vmax), ptr(pr.step),
                                                          pr.seed_int, counter, 0, ptr(p1), self._chain0))
                if self.isgood is not None:
                    self._redraw_with_rules(p, p1, counter, stream)
            c, status = self._solve_raw(p1)
            _lib.check(L.surfdisp_mcmc_accept_device(stream, C, N, int(self.periods.numel()), ptr(c), ptr(status),
                                                     ptr(st["c_obs"]), ptr(st["uncer"]), ptr(st["mask8"]),
                                                     1 if st["c_obs"].ndim == 2 else 0, ptr(p1), ptr(p), ptr(st["chi"]),
                                                     rowp, int(row_stride), pr.seed_int, counter, 1 if first else 0,
                                                     self._chain0))
        return p

    def _redraw_with_rules(self, p, p1, counter, stream):
        """The redraw loop of ``MCinv.perturb`` / ``reset`` (models.py:192-219) for a ``PriorRules`` predicate, on the device and
        without a host synchronisation: the prior kernel marks the chains whose proposal ``p1`` breaks a rule, the masked
        proposal kernel draws those again - ``rounds`` Gaussian rounds, ``reset_rounds`` uniform ones, then the chain's own
        state.  One tag per chain, rising from round to round (cleared once per step)."""
        import ctypes
        torch = self.torch
        rules = self.isgood
        C, N = p1.shape
        mb = rules.model
        idesc, fdesc, Lout = mb.native_descriptor()
        flags = rules.device_flags()
        st = self._fz
        if st.get("tags") is None or st["tags"].numel() != C:
            st["tags"] = torch.zeros(C, dtype=torch.uint8, device=self.device)
        tags = st["tags"]
        L = _lib.lib()
        ptr = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
        pr = self.proposer
        vmax = -1.0 if rules.vs_max is None else rules.vs_max
        pfull = lambda q: q if mb.n_aux == 0 else mb._full(q).contiguous()
        total = rules.rounds + rules.reset_rounds + 1
        tags.zero_()                                               # once per step: the rounds use rising tags 1, 2, ...
        for r in range(total):
            q = pfull(p1)
            # round r: chains redrawn in round r - 1 carry tag r (all chains in round 0); the ones that break a rule get r + 1 ...
            _lib.check(L.surfdisp_prior_device(stream, C, q.shape[1], int(Lout), ptr(q), ptr(idesc), ptr(fdesc), ptr(flags),
                                               ctypes.c_double(vmax), r if r > 0 else -1, r + 1, ptr(tags)))
            # ... and draw again: Gaussian rounds, then uniform prior draws, last the chain's own state
            mode = 0 if r < rules.rounds else (1 if r < rules.rounds + rules.reset_rounds else 2)
            _lib.check(L.surfdisp_mcmc_propose_masked_device(stream, C, N, ptr(p), ptr(pr.vmin), ptr(pr.vmax), ptr(pr.step),
                                                             pr.seed_int, counter, r, mode, ptr(tags), r + 1, ptr(p1), self._chain0))

    SPEC_MAX_STACKS = 2048          # stacks per speculative lock step: a 64-lane team = one wavefront each, the chip holds 4 096

    def auto_spec_depth(self, C):
        """Depth of the speculative lock step the fused path takes by default: the deepest tree (<= 4) whose
        C * (2^d - 1) stacks stay within ``SPEC_MAX_STACKS`` - a lock step of so few stacks is one wavefront per stack walking
        its periods one after the other on a mostly idle chip, and costs the same 0.9-1.0 ms for 100 stacks as for 700
        (100 chains x 96 layers: 0.96 ms per step plain, 0.345 with d = 3, 0.31 with d = 4 at 4.1 forward solves per step;
        scripts/time_speculative.py): 4 up to 136 chains, 3 up to 292, 2 up to 682, 1 from 683 chains on, and 1 where the (stack, period) decomposition fills the chip
        instead (``independent``: 0.24 ms per step plain, 0.29 with d = 3)."""
        if self.independent is True or (self.independent == "auto" and C < self.AUTO_INDEP_CHAINS):
            return 1
        if self.isgood is not None:
            return 1                                               # (prior rules: the redraw rounds belong to the plain lock step)
        for d in (4, 3, 2):
            if C * ((1 << d) - 1) <= self.SPEC_MAX_STACKS:
                return d
        return 1

    def fused_tree_step(self, p, depth, nsteps, row=None, row_stride=0, row_offset=0, step_stride=0, counter=None):
        """``nsteps`` <= ``depth`` Metropolis steps of every chain from ONE batched solve (speculative / "prefetching"
        Metropolis on the device: ``surfdisp_mcmc_propose_tree_device`` lays out the binary tree of the next ``depth``
        accept / reject outcomes, 2^depth - 1 proposals per chain, each drawn from the state its branch would be in; all go
        through one forward solve; ``surfdisp_mcmc_accept_tree_device`` walks the tree with the usual test).  Every
        proposal is drawn from and tested against the state the chain is in at that step: the chain is distributed exactly
        as with ``fused_step``.  mcTrack rows of the steps ``step_stride`` doubles apart, the first at ``row_offset``."""
        import ctypes
        torch = self.torch
        C, N = p.shape
        M = (1 << int(depth)) - 1
        st = self._fused_buffers(C)
        if st.get("M") != M:
            st["M"] = M
            st["q"] = torch.empty((C, M, N), dtype=torch.float64, device=self.device)
            st["qrows"] = torch.arange(C, device=self.device).repeat_interleave(M) if self.local_rows is not None else None
        L = _lib.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
        pr = self.proposer
        if counter is None:
            self._counter += 1
            counter = self._counter
        rowp = ctypes.c_void_p(row.data_ptr() + 8 * int(row_offset)) if row is not None else ctypes.c_void_p(0)
        with torch.cuda.device(self.device):
            _lib.check(L.surfdisp_mcmc_propose_tree_device(stream, C, N, int(depth), ptr(p), ptr(pr.vmin), ptr(pr.vmax), ptr(pr.step),
                                                           pr.seed_int, counter, ptr(st["q"]), self._chain0))
            c, status = self._solve_raw(st["q"].view(C * M, N), rows=st["qrows"])
            _lib.check(L.surfdisp_mcmc_accept_tree_device(stream, C, N, int(self.periods.numel()), int(depth), int(nsteps),
                                                          ptr(c), ptr(status), ptr(st["c_obs"]), ptr(st["uncer"]), ptr(st["mask8"]),
                                                          1 if st["c_obs"].ndim == 2 else 0, ptr(st["q"]), ptr(p), ptr(st["chi"]),
                                                          rowp, int(row_stride), int(step_stride), pr.seed_int, counter, self._chain0))
        return p

    # ------------------------------------------------------------------ chain groups
    GROUP_MIN_CHAINS = 4096         # fewer chains: one group (see chain_groups)

    def chain_groups(self, C, groups=None):
        """None, or a ``ChainGroups`` that advances the C chains as ``groups`` contiguous groups, each on its own stream.

        Chains are independent, so nothing orders one group's lock step against another's: while one group's root search
        drains (its last wavefronts leave SIMDs idle) and its small kernels run (proposal, stacks, prep, accept), the
        other group's root search has the chip - 25 600 chains x 96 layers: 6.3 -> 5.4 ms per step of all chains with two
        groups, 16 400 chains (one workgroup more than a full round of the root search): 5.1 -> 3.9 ms; never slower from
        1 600 chains on (three groups: as two; four lose 10-25 %; profiles/r03b/chain_groups.txt).  A group's solve
        carries SURFDISP_PIPELINED, so the library sizes its teams for both groups' stacks.  The random streams are
        indexed by the chain's index in the whole sampler (``chain0``), so every chain draws the same numbers for any number
        of groups: the chains - every mcTrack row - are then identical whenever the groups' launches use the lanes per stack
        the whole batch would (deep stacks: always 16 from 2 048 per launch on; tests/test_mcmc.py), and otherwise differ by
        what the solver's answers differ between team sizes (1e-6, which can flip a borderline accept).  Default: two groups from ``GROUP_MIN_CHAINS`` chains on (``PYSURFINV_CHAIN_GROUPS``
        overrides), one below."""
        if groups is None:
            env = os.environ.get("PYSURFINV_CHAIN_GROUPS")
            groups = int(env) if env else (2 if C >= self.GROUP_MIN_CHAINS else 1)
        groups = max(1, min(int(groups), C))
        if groups == 1 or not self.fused_available():
            return None
        key = (C, groups)
        if key not in self._groups:
            self._groups[key] = ChainGroups(self, C, groups)
        return self._groups[key]

    # ------------------------------------------------------------------ proposals
    def _good(self, p):
        if self.isgood is None:
            return self.torch.ones(p.shape[0], dtype=self.torch.bool, device=self.device)
        return self.isgood(p)

    def perturb(self, p):
        """MCinv.perturb: redraw the WHOLE proposal while isgood() rejects it (models.py:192-205)."""
        torch = self.torch
        new = self.proposer.move(p)
        if self.isgood is None:                                    # always good (models.py:220-224): no redraw loop,
            return new                                             # and no host synchronisation in the lock step
        bad = ~self._good(new)
        tries = 1
        while tries < 1000 and bool(bad.any()):
            idx = bad.nonzero(as_tuple=True)[0]
            new = new.clone(); new[idx] = self.proposer.move(p[idx])
            bad = ~self._good(new)
            tries += 1
        if bool(bad.any()):
            new = torch.where(bad[:, None], self.reset(p.shape[0]), new)
        return new

    def reset(self, C):
        """MCinv.reset: uniform prior draw, redrawn while isgood() rejects (models.py:206-219)."""
        new = self.proposer.reset(C)
        if self.isgood is None:
            return new
        bad = ~self._good(new)
        tries = 1
        while tries < 10000 and bool(bad.any()):
            idx = bad.nonzero(as_tuple=True)[0]
            new = new.clone(); new[idx] = self.proposer.reset(idx.numel())
            bad = ~self._good(new)
            tries += 1
        if bool(bad.any()):
            raise RuntimeError("Error: Cound not find a good model through reset.")   # models.py:219
        return new

    # ------------------------------------------------------------------ the sampler
    def run(self, n_chains, chainL, init_first=True, priori=False, _init_mask=None, spec_depth=None, fused=None, groups=None):
        """Advance ``n_chains`` chains for ``chainL`` steps each (= MCinvMP with runN = n_chains*chainL).

        Returns mcTrack float64 [n_chains, chainL, 3+N]; chain 0 starts at the initial model when
        ``init_first`` (point.py:48-51, MCinvMP passes init = (i==0), :97).

        spec_depth = d > 1: speculative ("prefetching") Metropolis for small chain counts, where one
        lock step is latency-bound and the GPU is mostly idle.  The binary tree of the next d
        accept/reject outcomes is laid out in advance (2^d - 1 proposals per chain, each drawn from
        the state its branch would be in), all of them go through ONE batched forward solve, and the
        chain then walks the tree with the usual accept rule: d Metropolis steps per lock step.
        Every proposal is still drawn from q(current state, .) and tested against the current state,
        so the chain is distributed exactly as with d = 1; only the order in which random numbers are
        consumed differs (the exact-replay path of the reference trace uses d = 1).  Default (None): on the fused device
        path ``auto_spec_depth`` (4 for up to 136 chains, 3 up to 292, 2 up to 682, else 1), on the torch path 1.

        ``fused`` (default: whenever ``fused_available()``): the lock step as device kernels around the solver
        (``fused_step``: Philox random numbers keyed by the proposer's seed; same proposal and accept distributions).
        ``groups``: chain groups of the fused path (``chain_groups``; every chain's random numbers do not depend on it)."""
        torch = self.torch
        C, N = int(n_chains), self.spec.n
        if fused is None:
            fused = self.fused_available()
        fused = bool(fused) and not priori                         # a priori run evaluates nothing: the torch loop below
        if fused and not self.fused_available():
            raise ValueError("fused=True needs the device proposer, no isgood callback (or a PriorRules one) and the HIP forward path")
        if spec_depth is None:
            spec_depth = self.auto_spec_depth(C) if (fused and groups in (None, 1)) else 1
        spec_depth = int(spec_depth)
        if spec_depth > 1 and groups not in (None, 1):
            raise ValueError("run: chain groups and speculative lock steps (spec_depth > 1) cannot be combined")
        if spec_depth > 1 and not priori and not fused:
            return self._run_speculative(n_chains, chainL, init_first, _init_mask, spec_depth)
        track = torch.zeros((C, chainL, 3 + N), dtype=torch.float64, device=self.device)
        if fused:
            # propose / accept kernels around the solver: mcTrack rows are written by the accept kernel itself
            p = self._start(C, init_first, _init_mask).contiguous().clone()
            stride = chainL * (3 + N)
            cg = self.chain_groups(C, groups) if spec_depth <= 1 else None
            base = self._counter
            if cg is None and spec_depth > 1:
                # speculative lock steps: the first row is the start model itself, then spec_depth steps per batched solve
                if spec_depth > 4:
                    raise ValueError("spec_depth of the fused path is at most 4")
                self.fused_step(p, row=track, row_stride=stride, row_offset=0, first=True, counter=base + 1)
                i, n = 1, 1
                while i < chainL:
                    ns = min(spec_depth, chainL - i)
                    n += 1
                    self.fused_tree_step(p, spec_depth, ns, row=track, row_stride=stride, row_offset=i * (3 + N),
                                         step_stride=3 + N, counter=base + n)
                    i += ns
                self._counter = base + n
                return track
            if cg is not None:
                cg.fork()
            for i in range(chainL):
                if cg is not None:
                    cg.step(p, row=track, row_stride=stride, row_offset=i * (3 + N), first=(i == 0), counter=base + i + 1)
                else:
                    self.fused_step(p, row=track, row_stride=stride, row_offset=i * (3 + N), first=(i == 0), counter=base + i + 1)
            if cg is not None:
                cg.join()
            self._counter = base + chainL
            return track
        p0 = self.reset(C) if not (init_first and C == 1) else None
        if init_first:
            v0 = torch.as_tensor(self.spec.v0, dtype=torch.float64, device=self.device)[None, :]
            if not bool(self._good(v0)[0]):
                v0 = self.perturb(v0)                              # point.py:50-51
            p0 = v0 if p0 is None else torch.cat([v0, p0[1:]], dim=0)
        if _init_mask is not None:                                # several points: chain 0 of each
            v0 = torch.as_tensor(self.spec.v0, dtype=torch.float64, device=self.device)[None, :]
            if not bool(self._good(v0)[0]):
                v0 = self.perturb(v0)
            p0 = torch.where(_init_mask[:, None], v0.expand_as(p0), p0)
        mis0, chi0, L0 = self.misfit(p0)
        track[:, 0, 0] = mis0; track[:, 0, 1] = L0; track[:, 0, 2] = 1.0; track[:, 0, 3:] = p0
        for i in range(1, chainL):
            p1 = self.perturb(p0)
            if priori:                                             # point.py:66-69
                track[:, i, 0] = 0.0; track[:, i, 1] = 1.0; track[:, i, 2] = 1.0; track[:, i, 3:] = p1
                p0 = p1
                continue
            mis1, chi1, L1 = self.misfit(p1)
            better = chi1 < chi0
            # the reference draws random() only when chi1 >= chi0 (point.py:35-37)
            u = torch.zeros_like(chi1)
            need = ~better
            if bool(need.any()):
                if C == 1:
                    u = self.proposer.uniform(1)
                else:
                    u = self.proposer.uniform(C)
            acc = better | (need & (u > 1.0 - torch.exp(-(chi1 - chi0) / 2.0)))
            track[:, i, 0] = mis1; track[:, i, 1] = L1; track[:, i, 2] = acc.to(torch.float64)
            track[:, i, 3:] = p1
            p0 = torch.where(acc[:, None], p1, p0)
            chi0 = torch.where(acc, chi1, chi0)
        return track

    def _start(self, C, init_first, _init_mask):
        torch = self.torch
        p0 = self.reset(C) if not (init_first and C == 1) else None
        if init_first or _init_mask is not None:
            v0 = torch.as_tensor(self.spec.v0, dtype=torch.float64, device=self.device)[None, :]
            if not bool(self._good(v0)[0]):
                v0 = self.perturb(v0)                              # point.py:50-51
            if _init_mask is not None:
                p0 = torch.where(_init_mask[:, None], v0.expand_as(p0), p0)
            else:
                p0 = v0 if p0 is None else torch.cat([v0, p0[1:]], dim=0)
        return p0

    def _run_speculative(self, n_chains, chainL, init_first, _init_mask, d):
        torch = self.torch
        C, N = int(n_chains), self.spec.n
        M = (1 << d) - 1                                           # proposals per chain per lock step
        track = torch.zeros((C, chainL, 3 + N), dtype=torch.float64, device=self.device)
        p = self._start(C, init_first, _init_mask)
        mis, chi, L = self.misfit(p)
        track[:, 0, 0] = mis; track[:, 0, 1] = L; track[:, 0, 2] = 1.0; track[:, 0, 3:] = p
        ar = torch.arange(C, device=self.device)
        i = 1
        while i < chainL:
            # lay out the tree: node k has children 2k+1 (accepted) and 2k+2 (rejected)
            S = torch.empty((C, 2 * M + 1, N), dtype=torch.float64, device=self.device)
            Q = torch.empty((C, M, N), dtype=torch.float64, device=self.device)
            S[:, 0] = p
            for lev in range(d):
                lo, hi = (1 << lev) - 1, (1 << (lev + 1)) - 1
                st = S[:, lo:hi].reshape(-1, N)
                q = self.perturb(st).reshape(C, hi - lo, N)
                Q[:, lo:hi] = q
                ks = torch.arange(lo, hi, device=self.device)
                S[:, 2 * ks + 1] = q
                S[:, 2 * ks + 2] = S[:, lo:hi]
            # ONE forward solve of C*M stacks; proposal m of chain i is held against chain i's observations
            misQ, chiQ, LQ = self.misfit(Q.reshape(-1, N), rows=ar.repeat_interleave(M) if (self.c_obs.ndim == 2 or self.local_rows is not None) else None)
            misQ, chiQ, LQ = misQ.reshape(C, M), chiQ.reshape(C, M), LQ.reshape(C, M)
            node = torch.zeros(C, dtype=torch.int64, device=self.device)
            for _ in range(min(d, chainL - i)):
                chi1, q = chiQ[ar, node], Q[ar, node]
                better = chi1 < chi
                u = self.proposer.uniform(C)
                acc = better | (~better & (u > 1.0 - torch.exp(-(chi1 - chi) / 2.0)))
                track[:, i, 0] = misQ[ar, node]; track[:, i, 1] = LQ[ar, node]
                track[:, i, 2] = acc.to(torch.float64); track[:, i, 3:] = q
                p = torch.where(acc[:, None], q, p)
                chi = torch.where(acc, chi1, chi)
                node = torch.where(acc, 2 * node + 1, 2 * node + 2)
                i += 1
        return track

    def run_points(self, n_points, chains_per_point, chainL, on_device=False, groups=None, spec_depth=None):
        """MCinvMP for n_points at once: chain index = point * chains_per_point + k; chain k = 0 of
        every point starts at the initial model, the others at prior draws (point.py:95-99).
        Returns float64 [n_points, chains_per_point, chainL, 3+N] (numpy, or the device tensor)."""
        torch = self.torch
        C = n_points * chains_per_point
        first = (torch.arange(C, device=self.device) % chains_per_point) == 0
        tr = self.run(C, chainL, init_first=False, _init_mask=first, groups=groups, spec_depth=spec_depth).reshape(n_points, chains_per_point, chainL, -1)
        return tr if on_device else tr.cpu().numpy()

    def summarise_points(self, track, obs_rows):
        """What the reference's ``PostPoint`` derives from one point's ``mcTrack`` (point.py:147-171), for all
        points at once on the device.  ``track`` [n_points, R, 3+N] (R = chains * chainL rows per point, chains one
        after the other as in the ``.npz``); ``obs_rows`` [n_points]: observation row of each point.

        * rejected rows take the parameters of the last accepted row before them (``trueMarkovChain``, :154-159);
        * ``minMod`` = row of the smallest misfit (:161-165); ``thres = max(2 min, min + 0.5)`` (:308-309);
        * ``avgMod`` = mean of the parameters of all rows with misfit < thres (:166-169), forward-solved once more
          for its misfit, likelihood (:171) and predicted curve (``pvelp``, model3D.py:32-35).
        Returns float64 [n_points, 6 + 2N + P]: min_misfit, min_L, thres, n_accepted_final, avg_misfit, avg_L,
        min_params[N], avg_params[N], pvelp[P]."""
        torch = self.torch
        npnt, R, W = track.shape
        N = W - 3
        acc = track[:, :, 2] > 0.5
        idx = torch.arange(R, device=track.device)[None, :].expand(npnt, R)
        last = torch.cummax(torch.where(acc, idx, torch.zeros_like(idx)), dim=1).values     # row 0 of a chain is accepted
        paras = torch.gather(track[:, :, 3:], 1, last[:, :, None].expand(npnt, R, N))
        mis = torch.nan_to_num(track[:, :, 0], nan=float("inf"))
        imin = mis.argmin(dim=1)
        ar = torch.arange(npnt, device=track.device)
        min_mis, min_L, min_par = mis[ar, imin], track[ar, imin, 1], paras[ar, imin]
        thres = torch.maximum(2.0 * min_mis, min_mis + 0.5)
        final = mis < thres[:, None]
        nfin = final.sum(dim=1)
        avg_par = (paras * final[:, :, None]).sum(dim=1) / nfin.clamp(min=1)[:, None]
        avg_mis, _, avg_L, cP = self.misfit(avg_par, rows=obs_rows, return_c=True)
        return torch.cat([min_mis[:, None], min_L[:, None], thres[:, None], nfin.to(torch.float64)[:, None],
                          avg_mis[:, None], avg_L[:, None], min_par, avg_par, cP], dim=1)

    def run_graphed(self, n_chains, chainL, init_first=True, rounds=24):
        """``run()`` with the whole Metropolis step - proposal, parameters -> stack, the forward
        kernels, misfit, accept/reject, bookkeeping - captured ONCE in a HIP graph and replayed
        ``chainL - 1`` times: at small chain counts the step is bound by the ~100 kernel launches of
        the torch glue, not by the solver.  Requirements: device proposer (TorchProposer), no
        ``isgood`` callback, a parameterisation with static structure (``Model1DBatch._static_sig``).
        The bound-rejection loop of ``BrownianVar.move`` (<= 1000 tries, brownian.py:20-27) becomes a
        fixed ``rounds`` redraws followed by the uniform fallback: a variable sitting exactly on a
        bound survives 24 redraws with probability 6e-8, so the proposal distribution is unchanged
        for all practical purposes.  Random numbers come from torch's default device generator
        (graph-safe Philox)."""
        torch = self.torch
        if self.isgood is not None or not self.fused_available():
            # (the captured step is torch glue with torch's graph-safe generator - the fused kernels take their Philox counter by
            # value, which a graph would freeze -, and that glue has no redraw loop: no predicate here, PriorRules or callback)
            raise ValueError("run_graphed needs the device proposer, no isgood predicate and the HIP forward path")
        C, N = int(n_chains), self.spec.n
        dev = self.device
        pr = self.proposer
        track = torch.zeros((C, chainL, 3 + N), dtype=torch.float64, device=dev)
        p = self._start(C, init_first, None)
        mis, chi, L = self.misfit(p)
        track[:, 0, 0] = mis; track[:, 0, 1] = L; track[:, 0, 2] = 1.0; track[:, 0, 3:] = p
        step = torch.ones(1, dtype=torch.int64, device=dev)          # row the next step writes
        p = p.clone(); chi = chi.clone()

        def one_step():
            # all redraw rounds at once: [C, rounds, N] candidates, take the first one inside the bounds
            draws = p[:, None, :] + pr.step * torch.randn((C, rounds, N), dtype=torch.float64, device=dev)
            ok = (draws < pr.vmax) & (draws > pr.vmin)
            first = ok.to(torch.int8).argmax(dim=1, keepdim=True)
            chosen = torch.gather(draws, 1, first).squeeze(1)
            uni = pr.vmin + (pr.vmax - pr.vmin) * torch.rand((C, N), dtype=torch.float64, device=dev)
            new = torch.where(ok.any(dim=1), chosen, uni)
            mis1, chi1, L1 = self.misfit(new)
            better = chi1 < chi
            u = torch.rand(C, dtype=torch.float64, device=dev)
            acc = better | (~better & (u > 1.0 - torch.exp(-(chi1 - chi) / 2.0)))
            row = torch.cat([mis1[:, None], L1[:, None], acc.to(torch.float64)[:, None], new], dim=1)
            track.index_copy_(1, step, row[:, None, :])
            p.copy_(torch.where(acc[:, None], new, p))
            chi.copy_(torch.where(acc, chi1, chi))
            step.add_(1)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                                # warm-up off the capture stream
            one_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if chainL > 2:                                              # rows 0 and 1 are written; capture
            g = torch.cuda.CUDAGraph()                              # records one_step without running it
            with torch.cuda.graph(g):
                one_step()
            for _ in range(chainL - 2):
                g.replay()
            self.n_forward += (chainL - 2) * C
        return track

    # ------------------------------------------------------------------ output (point.py:82-85,120-123)
    @staticmethod
    def save_npz(outdir, pid, mc_track, setting, obs, chainL):
        os.makedirs(outdir, exist_ok=True)
        mc = np.asarray(mc_track, dtype=np.float64).reshape(-1, np.asarray(mc_track).shape[-1])
        np.savez_compressed(f"{outdir}/{pid}.npz", mcTrack=mc, setting=dict(setting), obs=obs,
                            invMeta={"pid": pid, "chainL": chainL})
        return f"{outdir}/{pid}.npz"
