#!/usr/bin/env python3
"""bench.py -- layered-model dispersion forward solves/sec (20 periods) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): batch of 65 536 MCMC-perturbed 10-layer stacks per GPU,
Rayleigh phase + group velocity at 20 periods, synthetic inputs of SURVEY.md section 8(d).
A "step" = one pass of the hot path (prep + root-search + group-velocity kernels) over one
batch that is already resident in HBM.  The path shards over independent stacks: each rank
owns its own batch (weak scaling), no data-path collective; the only communication is the
barrier + MAX-reduce of the timing.

Prints ONE JSON line on rank 0 with metric/value plus:
  roofline     - dominant kernel (root search), algorithmic bytes per launch / HIP-event duration
  cpu_baseline - the reference Fortran (oracle/_ref, flang -O2) on one host core, bounded sample;
                 the OpenMP C port on all cores is reported beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 65536
NLAY = 10
NPER = 20
KIND = 2 | int(os.environ.get("BENCH_KIND_FLAGS", "0"), 0)   # Rayleigh (c + U); development: extra kind flags
ALG_BYTES_PER_SOLVE = 20 * NLAY + 8 * NPER    # 5 fp32 arrays in, (c, U) out = 360 B
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(per):
    """Bounded CPU sample on the host cores (rank 0, N=1 only).  Checker code, timed - never the product."""
    from pysurfinv_amd import synth
    from oracle import cport, refso
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, int(os.environ.get("SURFDISP_CPU_THREADS", "64"))))
    out = {}
    sample = synth.synth_models(8192, NLAY, seed=0)
    if refso.available():
        n = 6144                                     # ~9 s at ~700 solves/s/core
        t0 = time.perf_counter()
        refso.forward_batch(sample[:n, 0], sample[:n, 1], sample[:n, 2], sample[:n, 3], sample[:n, 4], per, KIND)
        dt = time.perf_counter() - t0
        out = {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "reference",
               "sample": f"first {n} of the bench batch (B=65536 L=10 P=20 Rayleigh c+U), "
                         "unmodified fast_surf Fortran built by oracle/build_ref.sh (flang -O2), "
                         "non-reentrant so one core"}
    # the reentrant C port, all host cores
    cport.forward_batch(sample[:256], per, KIND, nthreads=ncores)
    t0 = time.perf_counter()
    cport.forward_batch(sample, per, KIND, nthreads=ncores)
    dtp = time.perf_counter() - t0
    t0 = time.perf_counter()
    cport.forward_batch(sample[:2048], per, KIND, nthreads=1)
    dt1 = time.perf_counter() - t0
    port = {"value": sample.shape[0] / dtp, "unit": "solves/s", "cores": ncores, "kind": "port",
            "sample": "8192 stacks of the bench batch, oracle/surfdisp_oracle.c with OpenMP",
            "one_core_value": 2048 / dt1}
    if not out:
        out = dict(port)
    else:
        out["port_all_cores"] = port
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from pysurfinv_amd import _lib, forward, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL on ROCm.  BENCH_REHEARSAL=1 (development only): all ranks share cuda:0 and
        # rendezvous over gloo, to exercise this code path on a one-GPU box.
        rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
        if rehearsal:
            local_rank = 0
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if _lib.lib().surfdisp_device_count() < 1:
        raise SystemExit("no HIP device: the product path has no CPU fallback")

    per_np = synth.default_periods(NPER)
    model = torch.from_numpy(synth.synth_models(B_PER_GPU, NLAY, seed=rank)).to(dev)
    per = torch.from_numpy(per_np).to(dev)
    # Two batches in flight on two HIP streams: the root-search kernel's wavefronts finish at
    # different times, and a second independent batch fills the idle SIMD slots (measured +15 %).
    # Same work per step; the one-batch-in-flight rate is reported beside it.  With a second batch in
    # flight the launches carry SURFDISP_PIPELINED (lanes per stack chosen for both batches: 2 instead
    # of 4 here, +6 % measured); results are identical for every team size (tests/test_gpu_parity.py).
    NFLIGHT = int(os.environ.get("BENCH_IN_FLIGHT", "2"))
    plans = [forward.BatchPlan(B_PER_GPU, NLAY, NPER, device=dev) for _ in range(NFLIGHT)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NFLIGHT)]
    plan = plans[0]

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ring = forward.EventRing(args.steps)       # HIP events recorded on the launch stream, read after the sync

    def steps(n, nflight, record=False, fast_scan=False):
        for i in range(n):
            with torch.cuda.stream(streams[i % nflight]):
                plans[i % nflight].run(model, per, kind=KIND, events=ring.slot(i) if record else None,
                                       pipelined=nflight > 1, fast_scan=fast_scan)

    steps(max(args.warmup, NFLIGHT), NFLIGHT)
    barrier()
    t0 = time.perf_counter()
    steps(args.steps, 1, record=True)                      # one batch in flight (reported beside value);
    barrier()                                              # its kernels are bracketed by HIP events
    elapsed_one = time.perf_counter() - t0
    kms = ring.kernel_ms().mean(axis=0)                    # live: average over the K timed launches
    steps(args.warmup, NFLIGHT)
    barrier()
    t0 = time.perf_counter()
    steps(args.steps, NFLIGHT)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # beside the headline: the same workload with SURFDISP_EXACTSCAN (every scan grid point evaluated, as the
    # reference does; the default steps over certified intervals - bit-identical outputs, include/surfdisp.h)
    steps(args.warmup, NFLIGHT, fast_scan=True)
    barrier()
    t0 = time.perf_counter()
    steps(args.steps, NFLIGHT, fast_scan=True)
    barrier()
    elapsed_fast = time.perf_counter() - t0
    steps(1, 1)                                            # leave the default mode's results in the plans
    barrier()

    # every stack of every rank must have been solved (work was not skipped)
    n_ok = min(int((p.status == 0).sum().item()) for p in plans)
    ok = torch.tensor([n_ok], dtype=torch.int64, device=dev)
    if dist is not None:
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)


    if rank == 0:
        total_solves = world * B_PER_GPU * args.steps
        value = total_solves / elapsed
        phase_s = kms[1] * 1e-3
        achieved = ALG_BYTES_PER_SOLVE * B_PER_GPU / phase_s / 1e9
        traffic, valu = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("phase_kernel_hbm_bytes_per_launch")
                valu = tj.get("valu", {}).get("surfdisp_phase_kernel")
            except Exception:
                traffic = None
        recursion = None
        rfile = os.path.join(ROOT, "profiles", "recursion_ceiling.json")
        if os.path.exists(rfile):
            try:
                rj = json.load(open(rfile))
                ev = rj["evaluations_per_stack_default_scan"] * B_PER_GPU / phase_s
                recursion = {"achieved": ev, "peak": rj["ceiling_evaluations_per_s_no_divergence"],
                             "unit": "secular-function evaluations/s", "frac": ev / rj["ceiling_evaluations_per_s_no_divergence"],
                             "note": "secular-function evaluations of the default scan per stack (counted with an instrumented "
                                     "build, committed in profiles/recursion_ceiling.json) / live root-search kernel time, "
                                     "against the bare recursion's measured rate (scripts/microbench/issue_rate.hip)"}
            except Exception:
                recursion = None
        line = {
            "metric": "layered-model dispersion forward solves/sec (20 periods)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (root search, Love group velocity) + f64 (Rayleigh eigenfunction state)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 65536 MCMC-perturbed 10-layer stacks per GPU, "
                                   "Rayleigh phase+group velocity at 20 periods",
                       "stacks_per_gpu": B_PER_GPU, "layers": NLAY, "periods": NPER,
                       "wave": "Rayleigh c+U", "batches_in_flight": NFLIGHT,
                       "team_lanes": int(_lib.lib().surfdisp_get_team(B_PER_GPU * (2 if NFLIGHT > 1 else 1), NLAY)),
                       "team_lanes_one_batch_in_flight": int(_lib.lib().surfdisp_get_team(B_PER_GPU, NLAY)),
                       "sharding": f"independent stacks, {world} rank(s), no data-path collective"},
            "solved_fraction": float(ok.item()) / (world * B_PER_GPU),
            "value_one_batch_in_flight": world * B_PER_GPU * args.steps / elapsed_one,
            "value_fast_scan": B_PER_GPU * args.steps / elapsed_fast,         # this rank, opt-in SURFDISP_FASTSCAN
            "kernel_ms": {"prep": kms[0], "phase": kms[1], "group_and_finish": kms[2],
                          "how": "HIP events recorded on the launch stream around each kernel of the K timed "
                                 "one-batch-in-flight steps, read after the closing synchronisation"},
            "roofline": {"bound": "hbm", "kernel": "surfdisp_phase_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "valu_pmc": valu,
                         "recursion_ceiling": recursion,
                         "note": "algorithmic bytes = 360 B/solve x 65536 solves per launch; the path is "
                                 "VALU/transcendental-bound, not HBM-bound (SURVEY.md 8(d)); traffic and "
                                 "valu_pmc come from the committed rocprofv3 --pmc passes (profiles/)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(per_np)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
