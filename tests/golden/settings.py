"""The two model settings the driver golden vectors were captured with (data, shared by
make_golden_driver.py's CONT / OCEAN and the tests)."""
CONT = {
    'Sediment': {'H': [2., 'abs_pos', 1.5, 0.1], 'Vs': [[1.5, 'abs', 0.5, 0.05], [2.2, 'abs', 0.5, 0.05]]},
    'Crust': {'H': [35., 'abs', 10., 1.0],
              'Vs': [[3.4, 'abs', 0.3, 0.02], [3.6, 'abs', 0.3, 0.02], [3.8, 'abs', 0.3, 0.02], [3.9, 'abs', 0.3, 0.02]]},
    'Mantle': {'H': 160., 'Vs': [[4.4, 'abs', 0.4, 0.02], [4.35, 'abs', 0.4, 0.02], [4.4, 'abs', 0.4, 0.02],
                                 [4.5, 'abs', 0.4, 0.02], [4.6, 'abs', 0.4, 0.02]]},
    'Info': {'modelType': 'MCInv', 'refLayer': True},
}
OCEAN = {
    'OceanWater': {'H': 2.5},
    'OceanSediment': {'H': [0.4, 'rel_pos', 100, 0.05], 'Vs': [1.0, 0.5, 1.6, 0.05]},
    'OceanCrust': {'H': [7., 'abs', 2.5, 0.2], 'Vs': [[3.25, 'abs', 0.3, 0.02], [3.94, 'abs', 0.3, 0.02]]},
    'OceanMantle': {'BottomDepth': [200., 'abs', 30., 2.0],
                    'Vs': [[4.4, 'abs', 0.4, 0.02], [4.2, 'abs', 0.4, 0.02], [4.3, 'abs', 0.4, 0.02], [4.5, 'abs', 0.4, 0.02]]},
    'Info': {'modelType': 'MCInv', 'refLayer': False},
}
PERIODS = [8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 50, 60, 70, 80]
