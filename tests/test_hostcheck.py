"""CPU-side check of the group-velocity DEVICE math: tests/hostcheck compiles the very functions
the group kernel runs (layer_derive, drop_group, make_prop/prop_apply, rayleigh_sweep,
group_rayleigh, group_love - marked __host__ __device__ in surfdisp_kernels.hip) for the host with
hipcc, and this test compares them with the oracle on the golden cases, fed with the oracle's own
c and ellipticity.  No GPU needed; skipped if hipcc is absent."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HC = os.path.join(HERE, "hostcheck")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def hostlib():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    so = os.path.join(HC, "libhostcheck.so")
    if os.environ.get("SURFDISP_HOSTCHECK_LIB"):               # sanitizer build (scripts/sanitize_cpu.sh): use as is
        sys.path.insert(0, HC)
        import run_hostcheck
        return run_hostcheck
    src = [os.path.join(HC, "hostcheck.hip"), os.path.join(HERE, "..", "pysurfinv_amd", "csrc", "surfdisp_kernels.hip")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in src):
        subprocess.check_call([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
                               "-I" + os.path.join(HERE, "..", "include"),
                               "-I" + os.path.join(HERE, "..", "pysurfinv_amd", "csrc"),
                               "-shared", "-o", so, src[0]], stderr=subprocess.DEVNULL)
    sys.path.insert(0, HC)
    import run_hostcheck
    return run_hostcheck


@pytest.mark.parametrize("case", ["synth_L5_R", "synth_L10_R", "synth_L21_R", "synth_L64_R", "water_L9_R",
                                  "two_layer_R", "rough_L10_R", "synth_L10_L", "synth_L64_L", "water_L9_L"])
def test_group_velocity_device_math_on_host(hostlib, ref_cases, case):
    d = ref_cases[case]
    per = np.ascontiguousarray(d["periods"], np.float32)
    c, u, r = hostlib.oracle_dbg(d["model"], per, d["kind"])
    uh = hostlib.host_group(d["model"], per, d["kind"], c, r)
    ok = u != 0
    assert ok.any()
    assert np.abs(uh[ok] / u[ok] - 1).max() < 5e-6
