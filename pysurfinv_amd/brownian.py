"""Batched random-walk parameters with the semantics of the reference's ``brownian.py``.

Reference (``/root/reference/brownian.py``):
* ``BrownianVar(v, vmin, vmax, step)``: the step is clipped to ``|vmax-vmin|/2`` (``:7``);
  ``move()`` draws ``random.gauss(v, step)`` until ``vmin < vNew < vmax`` (strict), at most 1000
  tries, then falls back to ``reset()`` (``:20-27``); ``reset()`` is ``random.uniform(vmin, vmax)``
  (``:17-19``).
* ``BrownianVarMC(v, ref, width, type, step)``: bounds derived from ``ref``/``width``/``type``
  ('abs', 'abs_pos', 'rel', 'rel_pos', ``:44-63``), same move/reset.

Here N scalar parameters x B chains are moved at once on the device.  Two proposers:
* ``TorchProposer``   - vectorised rejection sampling with a ``torch.Generator`` (production);
* ``PythonRandomProposer`` - one chain, CPython's ``random`` consumed in exactly the reference's
  order (variable by variable, retry by retry): used to replay traces captured from the reference.
"""
from __future__ import annotations

import random as _pyrandom
from dataclasses import dataclass

import numpy as np


def bounds_from_mc(ref, width, kind):
    """vmin, vmax of a BrownianVarMC (brownian.py:44-63)."""
    if kind == "abs":
        return ref - width, ref + width
    if kind == "abs_pos":
        return max(ref - width, 0), max(ref + width, 0)
    if kind == "rel":
        return ref * (1 - width / 100), ref * (1 + width / 100)
    if kind == "rel_pos":
        return max(ref * (1 - width / 100), 0), max(ref * (1 + width / 100), 0)
    raise ValueError(f"unknown BrownianVarMC type {kind!r}")


@dataclass
class ParamSpec:
    """N random-walk scalars: start value, open interval (vmin, vmax), clipped step."""
    v0: np.ndarray
    vmin: np.ndarray
    vmax: np.ndarray
    step: np.ndarray
    names: list

    @staticmethod
    def from_entries(entries, names=None):
        """entries: list of ``[v, vmin, vmax, step]`` or ``[v, 'abs'|'abs_pos'|'rel'|'rel_pos',
        width, step]`` -- the two spellings ``layers.buildSeisLayer`` accepts (layers.py:583-598)."""
        v0, lo, hi, st = [], [], [], []
        for e in entries:
            v = float(e[0])
            if isinstance(e[1], str):
                a, b = bounds_from_mc(v, float(e[2]), e[1])
            else:
                a, b = float(e[1]), float(e[2])
            s = float(e[3])
            half = abs(b - a) / 2
            s = half if s > half else s                     # brownian.py:7 / :65-66
            v0.append(v); lo.append(a); hi.append(b); st.append(s)
        n = len(entries)
        return ParamSpec(np.array(v0), np.array(lo), np.array(hi), np.array(st),
                         list(names) if names is not None else [f"p{i}" for i in range(n)])

    @property
    def n(self):
        return self.v0.size


class TorchProposer:
    """Vectorised BrownianVar.move()/reset() for [B, N] parameter blocks on a torch device."""

    MAX_TRIES = 1000                                        # brownian.py:21

    def __init__(self, spec: ParamSpec, device, seed=None):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        f = lambda a: torch.as_tensor(a, dtype=torch.float64, device=self.device)
        self.vmin, self.vmax, self.step = f(spec.vmin), f(spec.vmax), f(spec.step)
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(int(seed))
        else:
            self.gen.seed()
        self.seed_int = int(self.gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF     # key of the fused device kernels' Philox stream

    def reset(self, B):
        torch = self.torch
        u = torch.rand((B, self.vmin.numel()), dtype=torch.float64, device=self.device, generator=self.gen)
        return self.vmin + (self.vmax - self.vmin) * u          # random.uniform(a, b) = a + (b-a)*random()

    def move(self, v, rounds=12):
        """Bounded Gaussian step.  The first ``rounds`` tries of every variable are drawn in one shot
        ([B, rounds, N] candidates, first one inside the bounds wins); the rare leftovers go through
        the reference's retry loop up to its 1000 tries, then the uniform fallback."""
        torch = self.torch
        draws = v[:, None, :] + self.step * torch.randn((v.shape[0], rounds, v.shape[1]), dtype=torch.float64,
                                                        device=self.device, generator=self.gen)
        ok = (draws < self.vmax) & (draws > self.vmin)
        first = ok.to(torch.int8).argmax(dim=1, keepdim=True)
        new = torch.gather(draws, 1, first).squeeze(1)
        bad = ~ok.any(dim=1)
        tries = rounds
        while tries < self.MAX_TRIES and bool(bad.any()):
            draw = v + self.step * torch.randn(v.shape, dtype=torch.float64, device=self.device, generator=self.gen)
            new = torch.where(bad, draw, new)
            bad = ~((new < self.vmax) & (new > self.vmin))
            tries += 1
        if bool(bad.any()):                                     # "No valid perturb, uniform reset instead!"
            new = torch.where(bad, self.reset(v.shape[0]), new)
        return new

    def uniform(self, B):
        return self.torch.rand(B, dtype=self.torch.float64, device=self.device, generator=self.gen)


class PythonRandomProposer:
    """Single-chain proposer that consumes CPython's ``random`` stream exactly like the reference:
    variables in order, each retried until it lands inside its bounds (brownian.py:20-27)."""

    def __init__(self, spec: ParamSpec, device="cpu", seed=None):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.spec = spec
        self.rng = _pyrandom
        self.rng.seed(seed)

    def _t(self, a):
        return self.torch.as_tensor(np.asarray(a)[None, :], dtype=self.torch.float64, device=self.device)

    def reset(self, B):
        assert B == 1
        return self._t([self.rng.uniform(a, b) for a, b in zip(self.spec.vmin, self.spec.vmax)])

    def move(self, v):
        assert v.shape[0] == 1
        cur = v[0].detach().cpu().numpy()
        out = []
        for x, a, b, s in zip(cur, self.spec.vmin, self.spec.vmax, self.spec.step):
            for _ in range(1000):
                nv = self.rng.gauss(float(x), float(s))
                if nv < b and nv > a:
                    break
            else:
                nv = self.rng.uniform(a, b)
            out.append(nv)
        return self._t(out)

    def uniform(self, B):
        assert B == 1
        return self.torch.as_tensor([self.rng.random()], dtype=self.torch.float64, device=self.device)
