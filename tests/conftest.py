import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_cases():
    """tests/golden/ref_cases.npz -> {case: dict(model, periods, kind, c, u)}."""
    z = np.load(os.path.join(GOLDEN, "ref_cases.npz"))
    cases = {}
    for key in z.files:
        case, field = key.split("/")
        if case == "__meta__":
            continue
        cases.setdefault(case, {})[field] = z[key]
    for d in cases.values():
        d["kind"] = int(d["kind"])
    return cases


def load_families():
    """tests/golden/ref_families.npz (one small fixture per soak family, tests/golden/make_golden_families.py)
    -> {case: dict(model, nlay, periods, kind, c, u)}."""
    z = np.load(os.path.join(GOLDEN, "ref_families.npz"))
    cases = {}
    for key in z.files:
        case, field = key.split("/")
        if case == "__meta__":
            continue
        cases.setdefault(case, {})[field] = z[key]
    for d in cases.values():
        d["kind"] = int(d["kind"])
    return cases


@pytest.fixture(scope="session")
def ref_cases():
    return load_cases()


@pytest.fixture(scope="session")
def ref_families():
    return load_families()


@pytest.fixture(scope="session")
def eus():
    return dict(np.load(os.path.join(GOLDEN, "test1_eus.npz")))


def relerr(x, ref):
    """max relative error over entries the reference solved; zeros must coincide."""
    x = np.asarray(x, np.float64); ref = np.asarray(ref, np.float64)
    ok = ref != 0
    if not ok.any():
        return 0.0
    return float(np.max(np.abs(x[ok] / ref[ok] - 1.0)))
