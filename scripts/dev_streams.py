#!/usr/bin/env python3
"""Do two batches in flight (two HIP streams, two plans) fill the root-search kernel's idle slots?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pysurfinv_amd import forward, synth
B, L, P = 65536, 10, 20
per = torch.from_numpy(synth.default_periods(P)).cuda()
model = torch.from_numpy(synth.synth_models(B, L, seed=0)).cuda()
for nstream in (1, 2, 3, 4):
    plans = [forward.BatchPlan(B, L, P) for _ in range(nstream)]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    K = 24
    def go():
        for i in range(K):
            with torch.cuda.stream(streams[i % nstream]):
                plans[i % nstream].run(model, per, kind=2)
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{nstream} stream(s): {dt/K*1e3:.3f} ms per batch, {B*K/dt/1e6:.2f} M solves/s", flush=True)
