"""Example model settings (the reference's setting dictionaries, ``models.py:42-51``) and the synthetic observations
used by ``bench.py``, ``scripts/run_grid.py`` and the tests - kept in the package so that no product entry script
depends on the benchmark module."""
from __future__ import annotations

# Continental model of the Metropolis legs: sediment + 4-coefficient crust + 5-coefficient mantle + reference mantle =
# 96 layers, 13 random-walk parameters (the setting the driver golden vectors were captured with,
# tests/golden/make_golden_driver.py)
MCMC_SETTING = {
    'Sediment': {'H': [2., 'abs_pos', 1.5, 0.1], 'Vs': [[1.5, 'abs', 0.5, 0.05], [2.2, 'abs', 0.5, 0.05]]},
    'Crust': {'H': [35., 'abs', 10., 1.0],
              'Vs': [[3.4, 'abs', 0.3, 0.02], [3.6, 'abs', 0.3, 0.02], [3.8, 'abs', 0.3, 0.02], [3.9, 'abs', 0.3, 0.02]]},
    'Mantle': {'H': 160., 'Vs': [[4.4, 'abs', 0.4, 0.02], [4.35, 'abs', 0.4, 0.02], [4.4, 'abs', 0.4, 0.02],
                                 [4.5, 'abs', 0.4, 0.02], [4.6, 'abs', 0.4, 0.02]]},
    'Info': {'modelType': 'MCInv', 'refLayer': True},
}
MCMC_PERIODS = [8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 50, 60, 70, 80]

# Oceanic model with a thermal mantle (BASELINE configs[4]): water + Cascadia sediment + crust + OceanMantleHybrid
C5_SETTING = {
    'OceanWater': {'H': 2.6},
    'OceanSedimentCascadia': {'H': [0.3, 'abs', 0.2, 0.03]},
    'OceanCrust': {'H': 4.4, 'Vs': [3.25, 3.94]},
    'OceanMantleHybrid': {'BottomDepth': 200, 'Conversion': 'Ritzwoller', 'ThermAge': [4, 'rel_pos', 200, 0.4],
                          'Vs': [[0, 'abs', 0.2, 0.01], [0, 'abs', 0.2, 0.01], [0, 'abs', 0.2, 0.01], [0, 'abs', 0.1, 0.01]]},
    'Info': {'modelType': 'MCInv', 'period': 10, 'refLayer': False},
}


def synthetic_observations(n_points, device, seed=100, setting=None, periods=None):
    """Model + synthetic per-point observations: the model's own curve at a random 'true' parameter vector per point
    (a prior draw shrunk towards the start model), 1 % uncertainty.  Returns (Model1DBatch, c_obs[n, P], uncer[n, P])."""
    import torch
    from .brownian import TorchProposer
    from .layers_batch import Model1DBatch
    setting = MCMC_SETTING if setting is None else setting
    periods = MCMC_PERIODS if periods is None else periods
    mb = Model1DBatch(setting, device=device)
    pr = TorchProposer(mb.spec, device, seed=seed)
    v0 = torch.as_tensor(mb.spec.v0, dtype=torch.float64, device=device)[None, :]
    truth = v0 + 0.3 * (pr.reset(n_points) - v0)
    c_true, st = mb.forward(truth, periods=periods)
    c0, _ = mb.forward(v0, periods=periods)
    c_true = torch.where((st != 0)[:, None] | (c_true < 0.01), c0.expand_as(c_true), c_true)   # unsolved draw: start model's curve
    c_obs = c_true.double().cpu().numpy()
    return mb, c_obs, 0.01 * c_obs
