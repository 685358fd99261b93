#!/usr/bin/env python3
"""GPU: does a HIP-graph replay survive rocprofv3's counter collection?  (r03: a `rocprofv3 --pmc` pass of the mcmc leg hung in
MetropolisBatch.run_graphed's replay; the leg's counter passes have skipped the graphed sampler since.)  Run under
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc SQ_WAVES -d <dir> -- python3 scripts/graph_pmc_probe.py torch|library
torch  : a graph of plain torch kernels only (nothing of this library) - separates the profiler from the library;
library: a graph holding one batched solve of this library (prep / root search / finish)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
dev = torch.device("cuda:0")
x = torch.ones(1 << 20, device=dev)
if mode == "torch":
    def body():
        for _ in range(4):
            x.mul_(1.0001).add_(1e-6)
else:
    import numpy as np
    from pysurfinv_amd import forward, synth, _lib
    m = torch.from_numpy(synth.synth_models(2048, 10, seed=0)).to(dev)
    per = torch.from_numpy(synth.default_periods(20)).to(dev)
    plan = forward.BatchPlan(2048, 10, 20, device=dev)
    def body():
        plan.run(m, per, kind=_lib.KIND_RAYLEIGH | _lib.PHASE_ONLY)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
print(mode, "warm-up done", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
print(mode, "captured", flush=True)
t0 = time.time()
for i in range(8):
    g.replay()
torch.cuda.synchronize()
print(mode, f"8 replays done in {time.time() - t0:.3f} s: OK", flush=True)
