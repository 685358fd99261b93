"""Settings the f-4 golden vectors (ref_therm.npz) were captured with (data, shared by
make_golden_therm.py and tests/test_thermseis.py).  Layer/parameter layout of the reference's own
commented Cascadia example (point.py:372-391), with the generic 'MCInv' model type."""
_HYB = {
    'OceanWater': {'H': 2.6},
    'OceanSedimentCascadia': {'H': [0.3, 'rel_pos', 100, 0.03]},
    'OceanCrust': {'H': 7, 'Vs': [3.25, 3.94]},
    'OceanMantleHybrid': {'BottomDepth': 200,
                          'Conversion': 'Ritzwoller',
                          'ThermAge': [4, 'rel_pos', 200, 0.4],
                          'Vs': [[0, 'abs', 0.4, 0.01], [0, 'abs', 0.4, 0.01],
                                 [0, 'abs', 0.4, 0.01], [0, 'abs', 0.2, 0.01]]},
    'Info': {'modelType': 'MCInv', 'period': 10, 'refLayer': True, 'lithoAgeQ': True, 'lithoAge': 3.0},
}


def _with(conv, info):
    import copy
    s = copy.deepcopy(_HYB)
    s['OceanMantleHybrid']['Conversion'] = conv
    s['Info'].update(info)
    return s


HYBRID_RITZ = _with('Ritzwoller', {})
HYBRID_YAMA = _with('Yamauchi', {'lithoAgeQ': False, 'refLayer': False})
HYBRID_YAMA['OceanMantleHybrid']['Tp'] = 1350
HYBRID_YAMA['OceanMantleHybrid']['ThermAge'] = [0.5, 0.0, 3.0, 0.1]      # young: melt starts at the top
# same model with a sediment prior bounded away from zero thickness: the layer structure is then the
# same for every draw (Model1DBatch._static_sig) and the Metropolis step is HIP-graph capturable
HYBRID_STATIC = _with('Ritzwoller', {})
HYBRID_STATIC['OceanSedimentCascadia']['H'] = [0.3, 'abs', 0.2, 0.03]
HYBRID_STATIC_YAMA = _with('Yamauchi', {'lithoAgeQ': False, 'refLayer': False})
HYBRID_STATIC_YAMA['OceanSedimentCascadia']['H'] = [0.3, 'abs', 0.2, 0.03]
HYBRID_STATIC_YAMA['OceanMantleHybrid']['Tp'] = [1350, 'abs', 40, 5]
HYBRID_STATIC_YAMA['OceanMantleHybrid']['ThermAge'] = [0.5, 0.0, 3.0, 0.1]
PERIODS = [10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 50, 60, 70, 80]
