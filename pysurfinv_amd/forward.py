"""Host side of the forward-call boundary (reference layer L2, models.py:11-33).

* ``_calForward`` mirrors ``pySurfInv.models._calForward`` (same name, arguments,
  return value and failure convention) on top of the HIP library.
* ``forward_batch`` / ``forward_batch_torch`` are the batched entry points the
  reference does not have: B layer stacks -> c[B,P], U[B,P] in one launch.

numpy / torch are plumbing only; the arithmetic is in libsurfdisp_hip.so.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from . import fast_surf as _fast_surf_mod


def _calForward(inProfile, wavetype="Ray", periods=(5, 10, 20, 40, 60, 80), debug=False):
    """models.py:11-33: profile rows (h, Vs, Vp, rho, qs, qp) -> cR[:nper] or None."""
    if wavetype == "Ray":
        ilvry = 2
    elif wavetype == "Love":
        ilvry = 1
    else:
        raise ValueError("Wrong surface wave type: %s!" % wavetype)
    inProfile = np.asarray(inProfile)
    ind = np.where(inProfile[0] > 1e-3)[0]                 # models.py:20
    h, Vs, Vp, rho, qs, qp = inProfile[:, ind]
    qsinv = 1.0 / qs
    nper = len(periods)
    per = np.zeros(200, dtype=np.float64)
    per[:nper] = periods[:]
    nlay = h.size
    (ur0, ul0, cr0, cl0) = _fast_surf_mod.fast_surf(nlay, ilvry, Vp, Vs, rho, h, qsinv, per, nper)
    if np.any(cr0[:nper] < 0.01):                          # models.py:29 (yes: cr0 for Love too)
        if debug:
            print(cr0[:nper])
        return None
    return cr0[:nper]


def forward_batch(model, periods, kind=2, nlay=None, device=0, independent=False, fast_scan=False, strict=False):
    """model float32 [B,5,L] rows (vp, vs, rho, h, qsinv) -> (c[B,P], u[B,P], status[B]).

    Host buffers in, host buffers out (C ABI surfdisp_forward_batch).  ``independent=True`` ORs
    SURFDISP_INDEPENDENT into ``kind`` (one team per (stack, period); see include/surfdisp.h).
    The scan evaluates every 0.01 km/s grid point as the reference does; ``fast_scan=True`` opts into the
    heuristic coarse-to-fine scan (SURFDISP_FASTSCAN); ``strict=True`` solves every stack with the kernel that restates
    the reference's arithmetic statement by statement (SURFDISP_STRICT: a verification mode, several times slower)."""
    if independent:
        kind = int(kind) | _lib.INDEPENDENT
    if strict:
        kind = int(kind) | _lib.STRICT
    if fast_scan:                                  # opt-in heuristic scan (SURFDISP_FASTSCAN); default: every grid point
        kind = int(kind) | _lib.FASTSCAN
    L = _lib.lib()
    model = np.ascontiguousarray(model, dtype=np.float32)
    if model.ndim != 3 or model.shape[1] != 5:
        raise ValueError("model must be [B, 5, L]")
    B, _, Lmax = model.shape
    per = np.ascontiguousarray(periods, dtype=np.float32).ravel()
    P = per.size
    c = np.zeros((B, P), np.float32); u = np.zeros((B, P), np.float32)
    status = np.zeros(B, np.int32)
    fp = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    ip = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    nl = None
    if nlay is not None:
        nlay = np.ascontiguousarray(nlay, dtype=np.int32)
        if nlay.size != B:
            raise ValueError("nlay must have B entries")
        nl = ip(nlay)
    _lib.check(L.surfdisp_forward_batch(int(device), B, Lmax, nl, fp(model), P, fp(per), int(kind),
                                        fp(c), fp(u), ip(status)))
    return c, u, status


class BatchPlan:
    """Device-resident, stream-ordered batched solve on torch tensors.

    Owns the workspace and output tensors for a fixed (B, L, P); ``run`` launches the
    three kernels on torch's current stream without allocating or synchronising, so it
    can be captured in a HIP graph."""

    def __init__(self, B, L, P, device="cuda:0"):
        import torch
        self.torch = torch
        self.B, self.L, self.P = int(B), int(L), int(P)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SurfdispError("BatchPlan needs a HIP device tensor (no CPU fallback)")
        lib = _lib.lib()
        self.ws_bytes = int(lib.surfdisp_workspace_bytes(self.B, self.L, self.P))
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        self.c = torch.zeros(self.B, self.P, dtype=torch.float32, device=self.device)
        self.u = torch.zeros(self.B, self.P, dtype=torch.float32, device=self.device)
        self.status = torch.zeros(self.B, dtype=torch.int32, device=self.device)

    def run_timed(self, model, periods, kind=2, nlay=None, independent=False):
        """As run(), but blocks and also returns the (prep, phase, group) kernel durations in ms,
        measured with HIP events on the launch stream (surfdisp_forward_batch_device_timed)."""
        return self.run(model, periods, kind=kind, nlay=nlay, _timed=True, independent=independent)

    def run(self, model, periods, kind=2, nlay=None, _timed=False, independent=False, events=None,
            pipelined=False, fast_scan=False, strict=False, want_ratio=False):
        """Launch the kernels on torch's current stream (no allocation, no sync).  ``events``: an
        ``EventRing`` slot (4 HIP events recorded on the launch stream around the kernels).
        ``pipelined``: the caller keeps a second batch in flight on another stream
        (``SURFDISP_PIPELINED``: a launch hint, same results)."""
        torch = self.torch
        if independent:
            kind = int(kind) | _lib.INDEPENDENT
        if pipelined:
            kind = int(kind) | _lib.PIPELINED
        if fast_scan:                                  # opt-in heuristic scan (SURFDISP_FASTSCAN)
            kind = int(kind) | _lib.FASTSCAN
        if strict:                                     # verification mode (SURFDISP_STRICT): the exact kernel for every stack
            kind = int(kind) | _lib.STRICT
        for t, shape in ((model, (self.B, 5, self.L)), (periods, (self.P,))):
            if (t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shape
                    or t.device != self.device):
                raise ValueError(f"expected contiguous float32 {shape} on {self.device}")
        if nlay is not None and (nlay.dtype != torch.int32 or nlay.numel() != self.B
                                 or nlay.device != self.device):
            raise ValueError("nlay must be int32 [B] on the same device")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._last_ws = self.workspace
        args = [ctypes.c_void_p(stream), self.B, self.L,
                ctypes.c_void_p(nlay.data_ptr() if nlay is not None else 0),
                ctypes.c_void_p(model.data_ptr()), self.P, ctypes.c_void_p(periods.data_ptr()),
                int(kind), ctypes.c_void_p(self.c.data_ptr()), ctypes.c_void_p(self.u.data_ptr()),
                ctypes.c_void_p(self.status.data_ptr()), ctypes.c_void_p(self.workspace.data_ptr()),
                self.ws_bytes]
        with torch.cuda.device(self.device):
            if _timed:
                ms = (ctypes.c_float * 3)()
                rc = _lib.lib().surfdisp_forward_batch_device_timed(*args, ms)
                _lib.check(rc)
                return self.c, self.u, self.status, tuple(float(x) for x in ms)
            if want_ratio and events is not None:
                raise ValueError("BatchPlan.run: want_ratio and events cannot be combined (no events variant of surfdisp_forward_batch_device2)")
            if want_ratio:
                # ABI 3: also the Rayleigh ellipticity (the reference's COMMON /o/ ratio, calcul.f:195) -> self.ratio [B, P]
                if getattr(self, "ratio", None) is None:
                    self.ratio = torch.zeros(self.B, self.P, dtype=torch.float32, device=self.device)
                a2 = args[:10] + [ctypes.c_void_p(self.ratio.data_ptr())] + args[10:]
                rc = _lib.lib().surfdisp_forward_batch_device2(*a2)
            elif events is not None:
                rc = _lib.lib().surfdisp_forward_batch_device_events(*args, events)
            else:
                rc = _lib.lib().surfdisp_forward_batch_device(*args)
        _lib.check(rc)
        if want_ratio:
            return self.c, self.u, self.status, self.ratio
        return self.c, self.u, self.status


    def fallback_count(self):
        """Stacks of the last ``run`` that the production root search handed to the exact fallback kernel
        (``surfdisp_workspace_fallback_count``; synchronises the current stream)."""
        n = ctypes.c_int(0)
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        with self.torch.cuda.device(self.device):
            _lib.check(_lib.lib().surfdisp_workspace_fallback_count(ctypes.c_void_p(stream), ctypes.c_void_p(getattr(self, '_last_ws', self.workspace).data_ptr()),
                                                                    self.B, self.L, self.P, ctypes.byref(n)))
        return int(n.value)

    def counters(self):
        """(stacks through the exact fallback kernel, brackets refined with NEVILL by the phase test, ellipticities evaluated
        again with the reference's arithmetic) of the last solve (``surfdisp_workspace_counters``; synchronises the stream)."""
        n = (ctypes.c_int * 3)()
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        with self.torch.cuda.device(self.device):
            _lib.check(_lib.lib().surfdisp_workspace_counters(ctypes.c_void_p(stream), ctypes.c_void_p(getattr(self, '_last_ws', self.workspace).data_ptr()),
                                                              self.B, self.L, self.P, n))
        return tuple(int(x) for x in n)

    def run_kernels(self, model, periods, kind=2, nlay=None, want_vp=True, want_rho=True, small_workspace=False):
        """Forward solve + analytic partial derivatives of the phase velocity
        (``surfdisp_forward_kernels_device``): returns (c, u, status, dcdb, dcda, dcdr) with the
        partials float32 [B, P, L] = d c(period) / d (Vs | Vp | rho) of input layer i (``None`` where
        not requested; Love has no dcda).  Same launch as ``run`` plus per-layer sums in the
        group-velocity kernel - what ``SensKernelPert`` obtains from 2L+1 perturbed solves."""
        torch = self.torch
        for t, shape in ((model, (self.B, 5, self.L)), (periods, (self.P,))):
            if (t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shape
                    or t.device != self.device):
                raise ValueError(f"expected contiguous float32 {shape} on {self.device}")
        if nlay is not None and (nlay.dtype != torch.int32 or nlay.numel() != self.B
                                 or nlay.device != self.device):
            raise ValueError("nlay must be int32 [B] on the same device")
        mk = lambda: torch.empty((self.B, self.P, self.L), dtype=torch.float32, device=self.device)
        dcdb = mk()
        dcda = mk() if (want_vp and (int(kind) & 3) == _lib.KIND_RAYLEIGH) else None
        dcdr = mk() if want_rho else None
        ptr = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if small_workspace:                            # the direct route into the rows (tests compare the two)
            ws, ws_bytes = self.workspace, self.ws_bytes
        else:
            if getattr(self, "kworkspace", None) is None:      # + the layer-major scratch of the partials, kept for reuse
                self.kws_bytes = int(_lib.lib().surfdisp_kernels_workspace_bytes(self.B, self.L, self.P))
                self.kworkspace = torch.empty(self.kws_bytes, dtype=torch.uint8, device=self.device)
            ws, ws_bytes = self.kworkspace, self.kws_bytes
        self._last_ws = ws
        with torch.cuda.device(self.device):
            rc = _lib.lib().surfdisp_forward_kernels_device(
                ctypes.c_void_p(stream), self.B, self.L, ptr(nlay), ptr(model), self.P, ptr(periods), int(kind),
                ptr(self.c), ptr(self.u), ptr(self.status), ptr(dcdb), ptr(dcda), ptr(dcdr),
                ptr(ws), ws_bytes)
        _lib.check(rc)
        return self.c, self.u, self.status, dcdb, dcda, dcdr


class EventRing:
    """n x 4 HIP events owned by the caller, recorded by ``BatchPlan.run(..., events=ring.slot(i))``
    on the launch stream; ``kernel_ms()`` (after the caller synchronised) returns the
    [n, 3] prep / root-search / group+finish durations in milliseconds."""

    def __init__(self, n):
        self.n = int(n)
        self._ev = (ctypes.c_void_p * (4 * self.n))()
        _lib.check(_lib.lib().surfdisp_events_create(4 * self.n, self._ev))

    def slot(self, i):
        return ctypes.cast(ctypes.byref(self._ev, 4 * (i % self.n) * ctypes.sizeof(ctypes.c_void_p)),
                           ctypes.POINTER(ctypes.c_void_p))

    def kernel_ms(self, used=None):
        out = np.zeros((self.n if used is None else used, 3))
        ms = ctypes.c_float()
        for i in range(out.shape[0]):
            for k in range(3):
                _lib.check(_lib.lib().surfdisp_events_elapsed_ms(self._ev[4 * i + k], self._ev[4 * i + k + 1],
                                                                 ctypes.byref(ms)))
                out[i, k] = ms.value
        return out

    def __del__(self):
        try:
            _lib.lib().surfdisp_events_destroy(4 * self.n, self._ev)
        except Exception:
            pass


def forward_batch_torch(model, periods, kind=2, nlay=None):
    """One-shot torch entry: allocates a BatchPlan and runs it (outputs stay on the device)."""
    B, _, L = model.shape
    plan = BatchPlan(B, L, periods.numel(), device=model.device)
    return plan.run(model, periods, kind=kind, nlay=nlay)


class JointPlan:
    """Rayleigh + Love, phase + group velocity of the same stacks (BASELINE configs[4] shape): two BatchPlans on two HIP
    streams.  ``order``: "concurrent" - both root searches share the chip from the start (each sized for half of it);
    "rayleigh_first" / "love_first" - the second wave type's kernels wait for the END OF THE FIRST ONE'S ROOT SEARCH
    (an event recorded between its kernels), so each root search has the whole chip and the first one's group-velocity
    kernel runs beside the second root search.  Outputs stay on the device: dict(cR, uR, cL, uL, statusR, statusL)."""

    def __init__(self, B, L, P, device="cuda:0", order=None):
        import os
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.order = order or os.environ.get("SURFDISP_JOINT_ORDER", "concurrent")
        if self.order not in ("concurrent", "concurrent_love", "rayleigh_first", "love_first"):
            raise ValueError("order: concurrent | concurrent_love | rayleigh_first | love_first")
        self.ray = BatchPlan(B, L, P, device=device)
        self.love = BatchPlan(B, L, P, device=device)
        self.s_ray = torch.cuda.Stream(device=self.device)
        self.s_love = torch.cuda.Stream(device=self.device)
        self._ring = EventRing(2)                   # own events when the caller brings none (ordering needs one)

    def run(self, model, periods, nlay=None, events=(None, None)):
        """``events``: (Rayleigh, Love) ``EventRing`` slots, recorded on each plan's own stream (measurement)."""
        torch = self.torch
        cur = torch.cuda.current_stream(self.device)
        for s in (self.s_ray, self.s_love):
            s.wait_stream(cur)                       # inputs were produced on the caller's stream
        evR = events[0] if events[0] is not None else self._ring.slot(0)
        evL = events[1] if events[1] is not None else self._ring.slot(1)
        conc = self.order in ("concurrent", "concurrent_love")      # (concurrent_love: the Love kernels are enqueued first)

        def ray():
            with torch.cuda.stream(self.s_ray):
                return self.ray.run(model, periods, kind=_lib.KIND_RAYLEIGH, nlay=nlay, pipelined=conc, events=evR)

        def love():
            with torch.cuda.stream(self.s_love):
                return self.love.run(model, periods, kind=_lib.KIND_LOVE, nlay=nlay, pipelined=conc, events=evL)

        def wait(stream, ev):                        # ev[2]: recorded after the root search of the other solve
            _lib.check(_lib.lib().surfdisp_stream_wait_event(ctypes.c_void_p(stream.cuda_stream), ev[2]))

        if self.order == "concurrent_love":
            cL, uL, sL = love()
            cR, uR, sR = ray()
        elif self.order == "love_first":
            cL, uL, sL = love()
            wait(self.s_ray, evL)
            cR, uR, sR = ray()
        else:
            cR, uR, sR = ray()
            if not conc:
                wait(self.s_love, evR)
            cL, uL, sL = love()
        cur.wait_stream(self.s_ray); cur.wait_stream(self.s_love)
        return dict(cR=cR, uR=uR, cL=cL, uL=uL, statusR=sR, statusL=sL)
