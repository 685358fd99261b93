#!/usr/bin/env python3
"""GPU box: the PCIe-inclusive rate of the host-buffer entry surfdisp_forward_batch (numpy in, numpy out) on the bench
batch and a few other sizes, beside raw pageable / pinned copy rates of the same byte counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysurfinv_amd import forward, synth
per = synth.default_periods(20)
for B, L in ((65536, 10), (16384, 10), (32768, 10), (49152, 10), (262144, 10), (16384, 64), (8192, 64), (25600, 96)):
    model = synth.synth_models(B, L, seed=0)
    forward.forward_batch(model, per, kind=2)
    t0 = time.perf_counter(); K = 5
    for _ in range(K):
        c, u, st = forward.forward_batch(model, per, kind=2)
    dt = (time.perf_counter() - t0) / K
    # the C call alone, output arrays allocated and touched beforehand
    import ctypes
    from pysurfinv_amd import _lib
    lib = _lib.lib(); fp = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    per32 = np.ascontiguousarray(per, np.float32)
    t0 = time.perf_counter()
    for _ in range(K):
        lib.surfdisp_forward_batch(0, B, L, None, fp(model), len(per32), fp(per32), 2, fp(c), fp(u), st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    dtc = (time.perf_counter() - t0) / K
    print(f"   C call alone, outputs preallocated: {dtc*1e3:.2f} ms = {B/dtc/1e6:.2f} M solves/s")
    nin, nout = model.nbytes, c.nbytes + u.nbytes + st.nbytes
    print(f"B={B} L={L}: {dt*1e3:.2f} ms per call = {B/dt/1e6:.2f} M solves/s (host buffers in and out; {nin/1e6:.1f} MB in, {nout/1e6:.1f} MB out)", flush=True)
x = torch.empty(13_107_200 // 4, dtype=torch.float32); xp = x.pin_memory(); d = torch.empty_like(x, device="cuda")
for name, src in (("pageable", x), ("pinned", xp)):
    d.copy_(src); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): d.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"H2D 13.1 MB {name}: {dt*1e3:.3f} ms = {13.1072/dt/1e3:.1f} GB/s")
    src.copy_(d); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): src.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"D2H 13.1 MB {name}: {dt*1e3:.3f} ms = {13.1072/dt/1e3:.1f} GB/s")
t0 = time.perf_counter()
for _ in range(10): xp.copy_(x)
dt = (time.perf_counter() - t0) / 10
print(f"host memcpy pageable -> pinned 13.1 MB: {dt*1e3:.3f} ms = {13.1072/dt/1e3:.1f} GB/s")
