"""Per-point local information the ref_local.npz vectors were captured with (data, shared by make_golden_local.py and
the tests): for each model the local keys (= columns) and one row per grid point."""
import copy

from settings import CONT

GAUSS = copy.deepcopy(CONT)
GAUSS['Crust']['Gauss'] = [[0.15, 'abs', 0.2, 0.02], 15.0, 4.0]      # A random walk, mu (per point below), sigma

LOCAL_TABLES = {
    # thermal oceanic model: topo (km; positive values move the stack's top up, models.py:74), lithoAge (Myr: Q age),
    # period (s: Q period), fixed water depth
    "hyb": (["topo", "lithoAge", "period", "OceanWater.H"],
            [[0.0, 3.0, 10.0, 2.6], [0.0, 0.8, 10.0, 2.9], [0.0, 9.0, 25.0, 2.2], [0.4, 3.0, 10.0, 2.6],
             [1.1, 5.5, 50.0, 3.1], [0.0, 1.5, 5.0, 2.45], [0.0, 12.0, 8.0, 3.4], [0.25, 6.0, 15.0, 1.8],
             [0.0, 2.2, 40.0, 2.75], [0.7, 4.4, 12.0, 2.05]]),
    # oceanic model with the mantle given by its bottom depth: only topo acts
    "ocean": (["topo"], [[0.0], [0.3], [1.2], [2.5], [0.05], [0.8], [1.9], [3.3]]),
    # Gaussian crustal anomaly: centre depth per point, and a per-point fixed sediment velocity
    "gauss": (["Crust.Gauss[1]", "Crust.Gauss[2]"],
              [[15.0, 4.0], [8.0, 2.5], [22.0, 6.0], [30.0, 3.0], [12.5, 5.0], [18.0, 1.5], [5.0, 4.5], [26.0, 8.0]]),
}
