"""CPU-only checks of the host logic and the C-ABI library (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from pysurfinv_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "surfdisp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(surfdisp_[a-z0-9_]+|fast_surf_)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.surfdisp_abi_version() == 4
    assert L.surfdisp_kernel_name(1) == b"surfdisp_phase_kernel"


def test_workspace_and_team_heuristics():
    from pysurfinv_amd import _lib
    L = _lib.lib()
    assert L.surfdisp_workspace_bytes(65536, 10, 20) >= 65536 * (9 * 10 + 20) * 4
    assert L.surfdisp_workspace_bytes(0, 10, 20) == 0
    L.surfdisp_set_team(0)
    os.environ.pop("SURFDISP_TEAM", None)
    assert L.surfdisp_get_team(1, 10) == 64              # one stack: a whole wavefront
    assert L.surfdisp_get_team(1 << 20, 10) == 4         # huge batch, alone on the chip: four lanes per stack ...
    assert L.surfdisp_get_team2(1 << 20, 10, 20, _lib.KIND_RAYLEIGH | _lib.PIPELINED) == 2     # ... two beside another batch in flight
    assert L.surfdisp_get_team2(1 << 20, 10, 20, _lib.KIND_LOVE) == 2                          # Love (certified coarse scan): two again
    assert L.surfdisp_get_team(65536, 10) == 4           # the bench workload
    g = L.surfdisp_get_team(1 << 20, 200)                # LDS bound forces wider teams
    assert g >= 8 and 4 * 200 * (256 // g) * 4 <= 80 * 1024
    # the workgroup's working stacks stay within 44 KB of LDS per 256 lanes (6 floats per layer + 24 per team; surfdisp_get_team
    # describes a Rayleigh c+U call), and the launch has at least 196 608 lanes - deep stacks beyond 16 lanes: 131 072
    for B, Lmax in ((65536, 10), (32768, 30), (65536, 30), (16384, 64), (8192, 64), (8192, 30), (25600, 96), (4096, 20), (100, 96)):
        g = L.surfdisp_get_team(B, Lmax)
        assert (6 * Lmax + 24) * (256 // g) * 4 <= 44 * 1024 or g == 64, (B, Lmax, g)
        assert B * g >= (131072 if (Lmax >= 24 and g >= 16) else 196608) or g == 64, (B, Lmax, g)
    assert L.surfdisp_get_team(32768, 30) == 8 and L.surfdisp_get_team(4096, 20) == 64
    assert L.surfdisp_get_team(8192, 64) == 16 and L.surfdisp_get_team(8192, 30) == 16 and L.surfdisp_get_team(16384, 30) == 16
    assert L.surfdisp_set_team(3) == _lib.ERR_INVALID
    assert L.surfdisp_set_team(16) == 0 and L.surfdisp_get_team(5, 5) == 16
    L.surfdisp_set_team(0)


def test_argument_errors_without_touching_a_gpu():
    from pysurfinv_amd import _lib, forward
    L = _lib.lib()
    with pytest.raises(ValueError):
        forward.forward_batch(np.zeros((4, 4, 10), np.float32), [10.0])
    with pytest.raises(_lib.SurfdispError):
        forward.forward_batch(np.ones((4, 5, 10), np.float32), np.arange(1, 300.0))   # P > 200
    with pytest.raises(_lib.SurfdispError):
        forward.forward_batch(np.ones((4, 5, 10), np.float32), [10.0], kind=3)
    assert b"invalid" in L.surfdisp_last_error()


def test_no_cpu_fallback_when_no_device():
    """On a box without a GPU the product path must fail loudly, not compute on the CPU."""
    from pysurfinv_amd import _lib, forward, synth
    if _lib.lib().surfdisp_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.SurfdispError):
        forward.forward_batch(synth.synth_models(2, 5), synth.default_periods(4))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pysurfinv_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no CPU fallback", ""), os.path.join(dp, f)


def test_synth_generator_is_deterministic_and_shaped():
    from pysurfinv_amd import synth
    a = synth.synth_models(8, 10, seed=0); b = synth.synth_models(8, 10, seed=0)
    assert a.shape == (8, 5, 10) and a.dtype == np.float32 and np.array_equal(a, b)
    assert (np.diff(a[:, 1], axis=1) >= 0).all()
    w = synth.water_models(3)
    assert (w[:, 1, 0] == 0).all()


def test_sediment_family_is_monotone_and_deterministic():
    """synth.sediment_models feeds the fast-scan GPU test and the scan soak: Vs and Vp never decrease with
    depth (otherwise the root search scans exactly and the test proves nothing), same seed -> same stacks."""
    from pysurfinv_amd import synth
    for kw in ({}, dict(max_layers=12), dict(water=True), dict(max_layers=12, water=True)):
        m = synth.sediment_models(256, 14, seed=7, **kw)
        assert m.shape == (256, 5, 14) and m.dtype == np.float32
        assert np.all(np.diff(m[:, 1, :], axis=1) >= 0) and np.all(np.diff(m[:, 0, :], axis=1) >= 0)
        assert np.all(m[:, 0, :] > 0) and np.all(m[:, 2, :] > 0) and np.all(m[:, 3, :] >= 0)
        assert np.all(m[:, 1, 1] < 3.0)                             # soft on top of the rock stack
        assert np.array_equal(m, synth.sediment_models(256, 14, seed=7, **kw))


def test_team_introspection_needs_no_device():
    """surfdisp_get_team / surfdisp_get_team2 are host logic (the sizing forward_device_impl applies): powers of two in
    1..64, wider for fewer stacks, Love never below 8 lanes, a phase-only Rayleigh launch of the grid leg's share at 16."""
    from pysurfinv_amd import _lib
    L = _lib.lib()
    R, LV, PH, PIPE, IND = _lib.KIND_RAYLEIGH, _lib.KIND_LOVE, _lib.PHASE_ONLY, _lib.PIPELINED, _lib.INDEPENDENT
    for B, Lmax in ((1, 10), (100, 96), (4096, 20), (65536, 10), (25600, 96), (16384, 64)):
        for kind in (R, R | PH, LV, R | PIPE, LV | PIPE, R | PH | IND):
            g = L.surfdisp_get_team2(B, Lmax, 20, kind)
            assert g in (1, 2, 4, 8, 16, 32, 64), (B, Lmax, kind, g)
    assert L.surfdisp_get_team2(1, 10, 20, R) == 64 and L.surfdisp_get_team2(65536, 10, 20, R) == 4
    assert L.surfdisp_get_team2(65536, 10, 20, R | PIPE) == 2           # three batches in flight: two-lane teams
    assert L.surfdisp_get_team2(25600, 96, 19, R | PH) == 16            # the grid leg (LDS: 8-lane teams would need 74 KB)
    assert L.surfdisp_get_team2(65536, 10, 20, LV) == 4 and L.surfdisp_get_team2(16384, 64, 20, LV | PIPE) == 16
    assert L.surfdisp_get_team2(100, 96, 19, R | PH | IND) <= L.surfdisp_get_team2(100, 96, 19, R | PH)


def test_committed_counter_profiles_are_of_these_sources():
    """profiles/traffic_{latest,grid,c5,mcmc}.json - what bench.py's roofline blocks read - carry the hash of the library
    sources they were measured on (hipcc's binaries are not bit-reproducible, the sources are): a kernel change without a
    new counter pass fails here instead of shipping a roofline of another build."""
    import json
    from pysurfinv_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    here = _lib.source_hash()
    assert len(here) == 16
    for leg in ("latest", "grid", "c5", "mcmc"):
        tj = json.load(open(os.path.join(root, "profiles", f"traffic_{leg}.json")))
        assert tj.get("src_sha256_16") == here, (leg, tj.get("src_sha256_16"), here)
