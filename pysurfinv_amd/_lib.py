"""ctypes binding of libsurfdisp_hip.so (C ABI: include/surfdisp.h).

There is no CPU fallback in this package: if the HIP library is missing, or no
gfx950 device is present, every compute entry point raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SURFDISP_LIB_PATH: developer override to load an experimental build of the same library (A/B measurements)
LIB_PATH = os.environ.get("SURFDISP_LIB_PATH") or os.path.join(_HERE, "lib", "libsurfdisp_hip.so")

SUCCESS, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_WORKSPACE = 0, -1, -2, -3, -4
OK, PARTIAL, NOROOT, BADMODEL, NUMERIC = 0, 1, 2, 4, 8
KIND_LOVE, KIND_RAYLEIGH = 1, 2
PHASE_ONLY = 0x10
INDEPENDENT = 0x20
PIPELINED = 0x40
EXACTSCAN = 0x80        # every grid point of the scan (Rayleigh: the default anyway; Love: switches the certified coarse scan off)
FASTSCAN = 0x100        # opt-in heuristic scan (include/surfdisp.h)
STRICT = 0x200          # verification mode: every stack through the statement-by-statement kernel
KERN_REFCOORD = 0x400   # surfdisp_forward_kernels_device: partials in the reference's coordinates (flattened, attenuated layer values)
NPER_MAX, NLAY_MAX = 200, 200

# every symbol include/surfdisp.h declares
EXPORTS = (
    "fast_surf_", "surfdisp_forward_batch", "surfdisp_thread_release", "surfdisp_workspace_bytes",
    "surfdisp_forward_batch_device", "surfdisp_forward_batch_device2", "surfdisp_forward_batch_device_timed",
    "surfdisp_forward_batch_device_events", "surfdisp_events_create", "surfdisp_events_destroy",
    "surfdisp_events_elapsed_ms", "surfdisp_stream_wait_event", "surfdisp_params_to_model_device",
    "surfdisp_params_to_model_thermal_device", "surfdisp_thermal_scratch_bytes",
    "surfdisp_mcmc_propose_device", "surfdisp_mcmc_accept_device", "surfdisp_prior_device", "surfdisp_mcmc_propose_masked_device", "surfdisp_mcmc_propose_tree_device", "surfdisp_mcmc_accept_tree_device",
    "surfdisp_forward_kernels_device", "surfdisp_kernels_workspace_bytes", "surfdisp_workspace_fallback_count", "surfdisp_workspace_counters", "surfdisp_set_team", "surfdisp_get_team", "surfdisp_get_team2",
    "surfdisp_device_count", "surfdisp_abi_version", "surfdisp_last_error",
    "surfdisp_kernel_name",
)


class SurfdispError(RuntimeError):
    pass


_lib = None


def lib() -> ctypes.CDLL:
    """Load the HIP library; raises SurfdispError (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SurfdispError(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). pysurfinv_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If this library
    # pulled in /opt/rocm's copy first, a later ``import torch`` would bring up a second runtime that
    # sees no GPU.  So when torch is installed let it load its runtime first; the dynamic linker then
    # binds libsurfdisp_hip.so's libamdhip64.so.N dependency to that same copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int)
    vp = ctypes.c_void_p
    L.fast_surf_.restype = None
    L.fast_surf_.argtypes = [ip, ip, fp, fp, fp, fp, fp, fp, ip, fp, fp, fp, fp]
    L.surfdisp_forward_batch.restype = ctypes.c_int
    L.surfdisp_forward_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, fp,
                                         ctypes.c_int, fp, ctypes.c_int, fp, fp, ip]
    L.surfdisp_workspace_bytes.restype = ctypes.c_size_t
    L.surfdisp_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.surfdisp_kernels_workspace_bytes.restype = ctypes.c_size_t
    L.surfdisp_kernels_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.surfdisp_forward_batch_device.restype = ctypes.c_int
    L.surfdisp_forward_batch_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp,
                                                ctypes.c_int, vp, ctypes.c_int, vp, vp, vp,
                                                vp, ctypes.c_size_t]
    L.surfdisp_forward_batch_device2.restype = ctypes.c_int
    L.surfdisp_forward_batch_device2.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp,
                                                 ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp,
                                                 vp, ctypes.c_size_t]
    L.surfdisp_forward_batch_device_timed.restype = ctypes.c_int
    L.surfdisp_forward_batch_device_timed.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp,
                                                      ctypes.c_int, vp, ctypes.c_int, vp, vp, vp,
                                                      vp, ctypes.c_size_t, fp]
    L.surfdisp_forward_batch_device_events.restype = ctypes.c_int
    L.surfdisp_forward_batch_device_events.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp,
                                                       ctypes.c_int, vp, ctypes.c_int, vp, vp, vp,
                                                       vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.surfdisp_events_create.restype = ctypes.c_int
    L.surfdisp_events_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.surfdisp_events_destroy.restype = ctypes.c_int
    L.surfdisp_events_destroy.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.surfdisp_stream_wait_event.restype = ctypes.c_int
    L.surfdisp_stream_wait_event.argtypes = [vp, vp]
    L.surfdisp_events_elapsed_ms.restype = ctypes.c_int
    L.surfdisp_events_elapsed_ms.argtypes = [vp, vp, fp]
    L.surfdisp_params_to_model_device.restype = ctypes.c_int
    L.surfdisp_params_to_model_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    L.surfdisp_forward_kernels_device.restype = ctypes.c_int
    L.surfdisp_forward_kernels_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, vp, ctypes.c_int,
                                                  vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t]
    L.surfdisp_thermal_scratch_bytes.restype = ctypes.c_size_t
    L.surfdisp_thermal_scratch_bytes.argtypes = [ctypes.c_int]
    L.surfdisp_params_to_model_thermal_device.restype = ctypes.c_int
    L.surfdisp_params_to_model_thermal_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp,
                                                          vp, ctypes.c_size_t, vp]
    u64 = ctypes.c_ulonglong
    L.surfdisp_mcmc_propose_device.restype = ctypes.c_int
    L.surfdisp_mcmc_propose_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, u64, u64, ctypes.c_int, vp, ctypes.c_long]
    if hasattr(L, "surfdisp_prior_device"):                    # (absent from an r03 build loaded through SURFDISP_LIB_PATH for A/B runs)
        L.surfdisp_mcmc_propose_masked_device.restype = ctypes.c_int
        L.surfdisp_mcmc_propose_masked_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, u64, u64, ctypes.c_int, ctypes.c_int,
                                                          vp, ctypes.c_int, vp, ctypes.c_long]
        L.surfdisp_prior_device.restype = ctypes.c_int
        L.surfdisp_prior_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_double, ctypes.c_int,
                                            ctypes.c_int, vp]
    L.surfdisp_mcmc_accept_device.restype = ctypes.c_int
    L.surfdisp_mcmc_accept_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int,
                                              vp, vp, vp, vp, ctypes.c_long, u64, u64, ctypes.c_int, ctypes.c_long]
    L.surfdisp_mcmc_propose_tree_device.restype = ctypes.c_int
    L.surfdisp_mcmc_propose_tree_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, u64, u64, vp, ctypes.c_long]
    L.surfdisp_mcmc_accept_tree_device.restype = ctypes.c_int
    L.surfdisp_mcmc_accept_tree_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp,
                                                   ctypes.c_int, vp, vp, vp, vp, ctypes.c_long, ctypes.c_long, u64, u64, ctypes.c_long]
    L.surfdisp_thread_release.restype = None
    L.surfdisp_thread_release.argtypes = []
    L.surfdisp_workspace_fallback_count.restype = ctypes.c_int
    L.surfdisp_workspace_fallback_count.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip]
    if hasattr(L, "surfdisp_workspace_counters"):              # (absent from an r03 build loaded through SURFDISP_LIB_PATH for A/B runs)
        L.surfdisp_workspace_counters.restype = ctypes.c_int
        L.surfdisp_workspace_counters.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip]
    L.surfdisp_set_team.restype = ctypes.c_int
    L.surfdisp_set_team.argtypes = [ctypes.c_int]
    L.surfdisp_get_team.restype = ctypes.c_int
    L.surfdisp_get_team.argtypes = [ctypes.c_int, ctypes.c_int]
    L.surfdisp_get_team2.restype = ctypes.c_int
    L.surfdisp_get_team2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.surfdisp_device_count.restype = ctypes.c_int
    L.surfdisp_abi_version.restype = ctypes.c_int
    L.surfdisp_last_error.restype = ctypes.c_char_p
    L.surfdisp_kernel_name.restype = ctypes.c_char_p
    L.surfdisp_kernel_name.argtypes = [ctypes.c_int]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != SUCCESS:
        msg = lib().surfdisp_last_error().decode(errors="replace")
        raise SurfdispError(f"libsurfdisp_hip error {rc}: {msg}")


def source_hash():
    """sha256 (16 hex digits) over the library's sources (csrc/*.hip, *.h, Makefile, include/surfdisp.h): what a counter
    profile under profiles/ is tagged with beside the binary's own hash - hipcc's output is not bit-reproducible, so a
    rebuilt library keeps the source tag while its binary hash changes."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h"))
                   + [os.path.join(here, "csrc", "Makefile"), os.path.join(os.path.dirname(here), "include", "surfdisp.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
