#!/usr/bin/env python3
"""bench.py -- layered-model dispersion forward solves/sec (20 periods) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload forward|grid|mcmc|c5]
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Default workload `forward` (BASELINE.json configs[1], the configuration the metric is quoted on): a batch of
65 536 MCMC-perturbed 10-layer stacks per GPU, Rayleigh phase + group velocity at 20 periods, synthetic inputs of
SURVEY.md section 8(d).  A "step" = one pass of the hot path (prep + root search [+ exact fallback] + group velocity
+ finish kernels) over one batch that is already resident in HBM.  The path shards over independent stacks: each rank
owns its own batch (weak scaling), no data-path collective; the only communication is the barrier + MAX-reduce of the
timing.  `value` is measured with the library's DEFAULT scan (every 0.01 km/s grid point, as the reference).

Prints ONE JSON line on rank 0 with metric / value plus
  roofline     - dominant kernel (root search): bound "valu"; achieved = wave-level VALU instructions per launch
                 (committed rocprofv3 --pmc pass of THIS library build, tagged) / live HIP-event duration, against
                 1024 SIMDs x one instruction per 2 cycles; the HBM figure north_star asks for is beside it
  cpu_baseline - the reference Fortran (oracle/_ref, flang -O2) on one host core, bounded sample; the OpenMP C port
                 on all cores beside it
  parity       - max relative error of c and U of the first 1024 stacks of the bench batch against the CPU oracle
                 (computed in the cpu_baseline leg, outside every timed region)
  workloads    - unless --workload names a single one: short legs of the other BASELINE configs on the same rank(s):
                 grid (configs[3] share: 512 points x 50 chains per GPU, lock-step Metropolis), mcmc (configs[2]),
                 c5 (configs[4]: joint R+L 64-layer thermal stacks + analytic sensitivity kernels)
"""
import argparse
import hashlib
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
SIMDS, CLOCK_HZ = 1024, 2.4e9 # 256 CUs x 4 SIMDs, max clock
VALU_PEAK = SIMDS * CLOCK_HZ / 2.0            # wave-level VALU instructions / s (one per 2 cycles per SIMD)

from pysurfinv_amd.settings import C5_SETTING, MCMC_PERIODS, MCMC_SETTING   # noqa: E402  (no torch import)


def lib_hash():
    from pysurfinv_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def src_hash():
    from pysurfinv_amd import _lib
    return _lib.source_hash()


def cpu_baseline(per, c_gpu, u_gpu):
    """Bounded CPU sample on the host cores (rank 0, N=1 only).  Checker code, timed - never the product.  Also the
    parity figure of the bench batch: its first 1024 stacks through the CPU oracle."""
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
        refso.forward_batch(sample[:n, 0], sample[:n, 1], sample[:n, 2], sample[:n, 3], sample[:n, 4], per, KIND & 3)
        dt = time.perf_counter() - t0
        out = {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "reference",
               "sample": f"first {n} of the bench batch (B=65536 L=10 P=20 Rayleigh c+U), "
                         "unmodified fast_surf Fortran built by oracle/build_ref.sh (flang -O2), "
                         "non-reentrant so one core"}
    # the reentrant C port, all host cores
    cport.forward_batch(sample[:256], per, KIND & 3, nthreads=ncores)
    t0 = time.perf_counter()
    co, uo, so = cport.forward_batch(sample, per, KIND & 3, nthreads=ncores)
    dtp = time.perf_counter() - t0
    t0 = time.perf_counter()
    cport.forward_batch(sample[:2048], per, KIND & 3, nthreads=1)
    dt1 = time.perf_counter() - t0
    port = {"value": sample.shape[0] / dtp, "unit": "solves/s", "cores": ncores, "kind": "port",
            "sample": "8192 stacks of the bench batch, oracle/surfdisp_oracle.c with OpenMP",
            "one_core_value": 2048 / dt1}
    if not out:
        out = dict(port)
    else:
        out["port_all_cores"] = port
    n = 1024
    rel = lambda x, r: float(np.max(np.abs(x[r != 0].astype(np.float64) / r[r != 0] - 1.0)))
    parity = {"max_rel_err_c": rel(c_gpu[:n], co[:n]), "max_rel_err_u": rel(u_gpu[:n], uo[:n]),
              "zero_pattern_equal": bool(np.array_equal(c_gpu[:n] > 0, co[:n] > 0)), "stacks": n, "bar": 1e-4,
              "against": "oracle/surfdisp_oracle.c (bit-exact restatement of the reference Fortran, tests/test_oracle.py), "
                         "first 1024 stacks of rank 0's bench batch, outside every timed region"}
    return out, parity


# ------------------------------------------------------------------------------------------------ helpers
def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process is only a launcher.  It
    starts N rank processes through torch.distributed.run (one per GPU, 127.0.0.1 rendezvous on a free port) as a CHILD
    process - no exec, and this parent never imports torch or touches HIP - passes their output through (rank 0
    prints the JSON line) and exits with the children's return code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"bench.py launcher: pid {os.getpid()} starts {n} ranks, torch imported in the launcher: {'torch' in sys.modules}",
          file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


class Runtime:
    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.local_rank = local_rank
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if args.gpus != self.world:
            raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={self.world}: launch one rank per GPU")
        # BENCH_REHEARSAL (development / tests only): "1" = all ranks share cuda:0 and rendezvous over gloo, to exercise the
        # N > 1 path on a one-GPU box; "cpu" = gloo and no device at all (only --workload launchcheck runs there)
        mode = os.environ.get("BENCH_REHEARSAL", "")
        self.rehearsal = mode in ("1", "cpu")
        self.cpu_only = mode == "cpu"
        self.dist = None
        self.backend = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.rehearsal:
                local_rank = 0
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            else:
                # nccl == RCCL on ROCm
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device(f"cuda:{local_rank}"))
            self.dist = dist
            self.backend = dist.get_backend()
        self.dev = torch.device("cpu") if self.cpu_only else torch.device(f"cuda:{local_rank}")
        if not self.cpu_only:
            torch.cuda.set_device(self.dev)

    @property
    def group_ranks(self):
        """World size as the process group itself reports it (RCCL ranks on a GPU node)."""
        return int(self.dist.get_world_size()) if self.dist is not None else 1

    def stamp(self, line):
        line.update({"rccl_ranks": self.group_ranks,
                     "collective_backend": ("rccl (torch.distributed nccl)" if self.backend == "nccl" else self.backend)})
        return line

    def barrier(self):
        if not self.cpu_only:
            self.torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
        if not self.cpu_only:
            self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, *vals):
        """MAX over ranks of each value (every timing of the line goes through here)."""
        if self.dist is None:
            return [float(v) for v in vals]
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=None if self.rehearsal else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def sum_over_ranks(self, *vals):
        if self.dist is None:
            return [int(v) for v in vals]
        t = self.torch.tensor(list(vals), dtype=self.torch.int64, device=None if self.rehearsal else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(x) for x in t.tolist()]

    def timed(self, fn, steps, warmup):
        """`warmup` untimed calls, then EXACTLY `steps` calls bracketed by barrier + synchronize on both sides;
        returns the MAX over ranks of the elapsed seconds."""
        for _ in range(warmup):
            fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)[0]


def leg_roofline(leg, live_ms, alg_bytes_per_launch, kernel_prefix="surfdisp_phase_kernel<2,", team=None, chip=None):
    """Roofline block of a side leg: its dominant kernel (the Rayleigh root search), VALU-bound.  achieved = wave-level
    VALU instructions per launch (rocprofv3 --pmc pass of THIS library build on this leg, profiles/traffic_<leg>.json,
    written by scripts/profile_leg.sh + scripts/summarise_leg.py) / the LIVE average duration of that kernel (HIP events
    on its stream inside this run); the HBM figure north_star asks for is beside it.  ``chip`` = (launches of this kernel
    per step, step duration in ms) when several launches share the chip (chain groups): the chip-level share beside the
    per-launch one, whose live duration includes the time the launch shares SIMDs with its neighbour."""
    live_s = live_ms * 1e-3
    hbm = alg_bytes_per_launch / live_s / 1e9 if live_s > 0 else None
    roof = {"bound": "valu", "kernel": None, "unit": "wave-level VALU instructions/s", "peak": VALU_PEAK,
            "achieved": None, "frac": None, "traffic": None, "live_kernel_ms": live_ms,
            "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS if hbm else None,
                    "algorithmic_bytes_per_launch": alg_bytes_per_launch}}
    tfile = os.path.join(ROOT, "profiles", f"traffic_{leg}.json")
    if not os.path.exists(tfile):
        roof["note"] = f"profiles/traffic_{leg}.json absent: no counter pass for this leg"
        return roof
    try:
        tj = json.load(open(tfile))
        # the default instantiation <KIND, G, INDEP = 0, FAST = 0, EXACT = 0> with the team size this leg's launches use
        cands = {k: v for k, v in tj["kernels"].items() if k.startswith(kernel_prefix) and "valu_wave_instructions" in v
                 and k.endswith(",0,0,0>") and (team is None or k.split(",")[1] == str(team))}
        if not cands:
            roof["note"] = f"no {kernel_prefix}..> instantiation" + (f" with teams of {team}" if team else "") + " in the profile"
            return roof
        name = max(cands, key=lambda k: cands[k].get("pct_of_gpu_time", 0.0))
        v = cands[name]
        here = lib_hash()
        roof.update({"kernel": name, "achieved": v["valu_wave_instructions"] / live_s,
                     "frac": v["valu_wave_instructions"] / live_s / VALU_PEAK,
                     "traffic": v.get("hbm_bytes_per_launch"),
                     "frac_measured_issue_costs": (v["valu_issue_frac_measured_costs"] * (v["kernel_cycles"] / CLOCK_HZ) / live_s
                                                   if "valu_issue_frac_measured_costs" in v else None),
                     "profile_kernel_avg_ms": v.get("avg_us", 0.0) / 1e3,
                     "lane_utilisation": v.get("lane_utilisation"), "mean_waves_per_simd": v.get("mean_waves_per_simd"),
                     "pmc_profile": tj.get("round"), "pmc_lib_sha256_16": tj.get("lib_sha256_16"), "this_lib_sha256_16": here,
                     "pmc_matches_this_build": tj.get("lib_sha256_16") == here,
                     "pmc_matches_this_source": tj.get("src_sha256_16") == src_hash()})
        if chip is not None and chip[1] > 0:
            n, step_s = int(chip[0]), chip[1] * 1e-3
            issue = (v["valu_issue_frac_measured_costs"] * (v["kernel_cycles"] / CLOCK_HZ) if "valu_issue_frac_measured_costs" in v else None)
            roof["chip_level"] = {"launches_per_step": n, "step_ms": chip[1],
                                  "frac": n * v["valu_wave_instructions"] / step_s / VALU_PEAK,
                                  "frac_measured_issue_costs": n * issue / step_s if issue is not None else None,
                                  "note": "all launches of this kernel in one step / the step's duration (the small kernels "
                                          "of the step included in the time, not in the instructions)"}
    except Exception as e:
        roof["note"] = f"profiles/traffic_{leg}.json unreadable: {e}"
    return roof


def chip_level_share(leg, kernels, step_ms):
    """VALU issue share of a step in which several kernels share the chip (the two root searches, group-velocity and
    ellipticity kernels of a joint solve): the named kernels' instructions (the leg's PMC pass) / the step's duration."""
    tfile = os.path.join(ROOT, "profiles", f"traffic_{leg}.json")
    try:
        tj = json.load(open(tfile))["kernels"]
        used = {n: tj[n] for n in kernels if n in tj and "valu_wave_instructions" in tj[n]}
        if not used or step_ms <= 0:
            return None
        step_s = step_ms * 1e-3
        instr = sum(v["valu_wave_instructions"] for v in used.values())
        issue = sum(v["valu_issue_frac_measured_costs"] * v["kernel_cycles"] / CLOCK_HZ for v in used.values()
                    if "valu_issue_frac_measured_costs" in v)
        return {"kernels": sorted(used), "step_ms": step_ms, "frac": instr / step_s / VALU_PEAK,
                "frac_measured_issue_costs": issue / step_s,
                "note": "VALU wave-instructions of the listed kernels (one launch each per step) / the step's duration"}
    except Exception as e:
        return {"note": f"profiles/traffic_{leg}.json unreadable: {e}"}


# ------------------------------------------------------------------------------------------------ workloads
def workload_forward(rt, args):
    torch = rt.torch
    from pysurfinv_amd import _lib, forward, synth
    dev, world, rank = rt.dev, rt.world, rt.rank
    per_np = synth.default_periods(NPER)
    model = torch.from_numpy(synth.synth_models(B_PER_GPU, NLAY, seed=rank)).to(dev)
    per = torch.from_numpy(per_np).to(dev)
    # Three batches in flight on three HIP streams: one batch is ONE wave of work for the chip (65 536 teams, fewer
    # than the lanes it holds), its wavefronts finish at different times, and the next independent batches fill the
    # SIMD slots it vacates (measured, scripts/sweep_inflight.sh: 26.8 / 28.8 / 30.9 M solves/s with 1 / 2 / 3 in
    # flight).  Same work per step; the one-batch-in-flight rate is reported beside it.  With more than one batch in
    # flight the launches carry SURFDISP_PIPELINED (lanes per stack chosen for the batches together); results are
    # identical for every team size (tests/test_gpu_parity.py).
    NFLIGHT = int(os.environ.get("BENCH_IN_FLIGHT", "3"))
    HINT = os.environ.get("BENCH_PIPELINED_HINT", "1") == "1"         # development: launch without the hint
    plans = [forward.BatchPlan(B_PER_GPU, NLAY, NPER, device=dev) for _ in range(NFLIGHT)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NFLIGHT)]
    ring = forward.EventRing(args.steps)       # HIP events recorded on the launch stream, read after the sync

    def steps(n, nflight, record=False, fast_scan=False):
        for i in range(n):
            with torch.cuda.stream(streams[i % nflight]):
                plans[i % nflight].run(model, per, kind=KIND, events=ring.slot(i) if record else None,
                                       pipelined=(nflight > 1) and HINT, fast_scan=fast_scan)

    steps(max(args.warmup, NFLIGHT), NFLIGHT)
    rt.barrier()
    t0 = time.perf_counter()
    steps(args.steps, 1, record=True)                      # one batch in flight (reported beside value);
    rt.barrier()                                           # its kernels are bracketed by HIP events
    elapsed_one = time.perf_counter() - t0
    kms = ring.kernel_ms().mean(axis=0)                    # live: average over the K timed launches
    steps(args.warmup, NFLIGHT)
    rt.barrier()
    t0 = time.perf_counter()
    steps(args.steps, NFLIGHT)                             # THE timed region of `value`
    rt.barrier()
    elapsed = time.perf_counter() - t0
    # beside the headline: the same workload with the opt-in count-guided scan (SURFDISP_FASTSCAN, include/surfdisp.h)
    steps(args.warmup, NFLIGHT, fast_scan=True)
    rt.barrier()
    t0 = time.perf_counter()
    steps(args.steps, NFLIGHT, fast_scan=True)
    rt.barrier()
    elapsed_fast = time.perf_counter() - t0
    elapsed, elapsed_one, elapsed_fast = rt.max_over_ranks(elapsed, elapsed_one, elapsed_fast)
    steps(1, 1)                                            # leave the default mode's results in plan 0
    rt.barrier()
    # every stack of every rank must have been solved (work was not skipped); none took the exact fallback
    n_ok, n_fb = rt.sum_over_ranks(min(int((p.status == 0).sum().item()) for p in plans), plans[0].fallback_count())

    if rank != 0:
        return None
    phase_s = kms[1] * 1e-3
    hbm_achieved = ALG_BYTES_PER_SOLVE * B_PER_GPU / phase_s / 1e9
    roof = {"bound": "valu", "kernel": "surfdisp_phase_kernel", "unit": "wave-level VALU instructions/s",
            "peak": VALU_PEAK, "achieved": None, "frac": None, "traffic": None,
            "hbm": {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
                    "note": "algorithmic bytes = 360 B/solve x 65536 solves per launch / live kernel duration: the "
                            "figure north_star asks for; the path is not HBM-bound (SURVEY.md 8(d))"}}
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    here = lib_hash()
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            v = tj.get("valu", {}).get("surfdisp_phase_kernel", {})
            same = tj.get("lib_sha256_16") == here
            roof.update({"achieved": v["valu_wave_instructions"] / phase_s,
                         "frac": v["valu_wave_instructions"] / phase_s / VALU_PEAK,
                         "traffic": tj.get("phase_kernel_hbm_bytes_per_launch"),
                         "frac_measured_issue_costs": (v["valu_issue_frac_measured_costs"] * (v["kernel_cycles"] / CLOCK_HZ) / phase_s
                                                       if "valu_issue_frac_measured_costs" in v else None),
                         "valu_pmc": v, "pmc_profile": tj.get("round"), "pmc_lib_sha256_16": tj.get("lib_sha256_16"),
                         "this_lib_sha256_16": here, "pmc_matches_this_build": bool(same),
                         "pmc_matches_this_source": tj.get("src_sha256_16") == src_hash(),   # (a rebuilt binary has another hash, the same sources)
                         "note": "achieved = wave-level VALU instructions per launch (SQ_INSTS_VALU of the committed "
                                 "rocprofv3 --pmc pass named in pmc_profile) / the LIVE duration of the kernel (HIP "
                                 "events on its stream); peak = 1024 SIMDs x 2.4 GHz / 2 cycles per instruction; "
                                 "frac_measured_issue_costs prices the same stream with the costs measured on this chip "
                                 "(scripts/microbench/valu_rates.hip: plain fp32 2.2, transcendental 8.1, fp64 4.2 SIMD "
                                 "cycles per wave instruction) - the share of the SIMDs' issue time the kernel fills; "
                                 "traffic = HBM bytes per launch from the same passes (FETCH_SIZE x 2 + WRITE_SIZE)"
                                 + ("" if same else "; WARNING: the profile was taken with another build of the library")})
        except Exception as e:                                   # a broken profile file must not kill the line
            roof["note"] = f"profiles/traffic_latest.json unreadable: {e}"
    line = {
        "metric": "layered-model dispersion forward solves/sec (20 periods)",
        "value": world * B_PER_GPU * args.steps / elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (root search, Love group velocity) + f64 (Rayleigh eigenfunction state)",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 65536 MCMC-perturbed 10-layer stacks per GPU, "
                               "Rayleigh phase+group velocity at 20 periods",
                   "stacks_per_gpu": B_PER_GPU, "layers": NLAY, "periods": NPER,
                   "wave": "Rayleigh c+U", "scan": "default: every 0.01 km/s grid point, as the reference",
                   "batches_in_flight": NFLIGHT,
                   "team_lanes": int(_lib.lib().surfdisp_get_team2(B_PER_GPU, NLAY, NPER, _lib.KIND_RAYLEIGH | (_lib.PIPELINED if (NFLIGHT > 1 and HINT) else 0))),
                   "team_lanes_one_batch_in_flight": int(_lib.lib().surfdisp_get_team2(B_PER_GPU, NLAY, NPER, _lib.KIND_RAYLEIGH)),
                   "sharding": f"independent stacks, {world} rank(s), no data-path collective"},
        "solved_fraction": n_ok / (world * B_PER_GPU),
        "stacks_through_exact_fallback": n_fb,
        "value_one_batch_in_flight": world * B_PER_GPU * args.steps / elapsed_one,
        "value_fast_scan": world * B_PER_GPU * args.steps / elapsed_fast,
        "fast_scan": "opt-in SURFDISP_FASTSCAN, NOT the headline: coarse scan guided by an exact count of the mode branches below each "
                     "trial (Wittrick-Williams, carried by the recursion); equal counts do not exclude a branch crossed twice at a "
                     "zero-group-velocity point - 2 of 1.2e9 soak stacks differed from the point-by-point scan (DESIGN.md section 10)",
        "kernel_ms": {"prep": kms[0], "phase": kms[1], "group_and_finish": kms[2],
                      "how": "HIP events recorded on the launch stream around each kernel of the K timed "
                             "one-batch-in-flight steps, read after the closing synchronisation; phase = root search + "
                             "the (idle) exact fallback launch behind it; group_and_finish = ellipticity kernel + "
                             "group-velocity kernel + finish"},
        "roofline": roof,
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"], line["parity"] = cpu_baseline(per_np, plans[0].c.cpu().numpy(), plans[0].u.cpu().numpy())
    return line


def _mcmc_setup(rt, n_points, chains):
    """Continental model + synthetic per-point observations (pysurfinv_amd.settings.synthetic_observations)."""
    from pysurfinv_amd.settings import synthetic_observations
    return synthetic_observations(n_points, rt.dev, seed=100 + rt.rank)


def workload_grid(rt, args, steps=None, warmup=None):
    """BASELINE configs[3], one lock step = one Metropolis step of every chain of every point a rank owns (512 points
    x 50 chains = 25 600 chains per GPU; 4 096 points on 8 GPUs): parameters -> 96-layer stacks (HIP), one batched
    phase-only forward solve, misfit + accept (torch).  Weak scaling over points; no data-path collective."""
    torch = rt.torch
    from pysurfinv_amd.mcmc import MetropolisBatch
    pts, chains = int(os.environ.get("BENCH_GRID_POINTS", "512")), int(os.environ.get("BENCH_GRID_CHAINS", "50"))
    K = steps if steps is not None else args.steps
    W = warmup if warmup is not None else args.warmup
    rep = lambda a: np.repeat(a, chains, axis=0)
    total = int(getattr(args, "total_points", 0) or 0)
    if total:
        # strong scaling: one grid for any rank count.  Every rank derives the observations of the WHOLE grid from the same
        # seed and keeps its block; its chains carry their index in the whole sampler (chain0), which keys the random streams
        from pysurfinv_amd.settings import synthetic_observations
        from pysurfinv_amd.shard import shard_range
        lo, hi = shard_range(total, rt.rank, rt.world)
        pts = hi - lo
        mb, c_all, unc_all = synthetic_observations(total, rt.dev, seed=100)
        c_obs, unc = c_all[lo:hi], unc_all[lo:hi]
        mc = MetropolisBatch(mb.spec, mb.to_model, MCMC_PERIODS, rep(c_obs), rep(unc), device=rt.dev, seed=7)
        mc._chain0 = lo * chains
    else:
        mb, c_obs, unc = _mcmc_setup(rt, pts, chains)
        mc = MetropolisBatch(mb.spec, mb.to_model, MCMC_PERIODS, rep(c_obs), rep(unc), device=rt.dev, seed=7 + rt.rank)
    C = pts * chains
    if os.environ.get("BENCH_GRID_PRIOR") == "1":
        # the lock step with prior predicates on the device (PriorRules: Vs increasing in sediment and crust, no drop across group
        # boundaries, Vs <= 4.9 km/s - the generic tests of the reference's model classes, models.py:294-320)
        from pysurfinv_amd.mcmc import PriorRules
        mc.isgood = PriorRules(mb, vs_max=4.9)
    state = {}

    fused = mc.fused_available() and os.environ.get("BENCH_GRID_FUSED", "1") == "1"
    track_row = torch.zeros((C, 3 + mb.spec.n), dtype=torch.float64, device=rt.dev)

    # chain groups as the library forms them (MetropolisBatch.chain_groups: two from 4 096 chains on, each on its own
    # stream; BENCH_GRID_GROUPS overrides) - every chain draws the same random numbers however they are grouped
    genv = os.environ.get("BENCH_GRID_GROUPS")
    cg = mc.chain_groups(C, int(genv) if genv else None) if fused else None

    def setup():
        state["p"] = mc.reset(C).contiguous()
        if cg is not None:
            cg.fork()
            cg.step(state["p"], first=True)
        elif fused:
            mc.fused_step(state["p"], first=True)          # chi-square of the start models into the sampler's state
        else:
            state["chi"] = mc.misfit(state["p"])[1]

    def step_fused():
        # the step of every chain as the library runs it (MetropolisBatch.run): propose kernel, parameters -> stacks,
        # prep / root search / finish, accept kernel (misfit, accept rule, state update, mcTrack row), per chain group
        if cg is not None:
            cg.step(state["p"], row=track_row, row_stride=3 + mb.spec.n)
        else:
            mc.fused_step(state["p"], row=track_row, row_stride=3 + mb.spec.n)

    def step():
        if fused:
            return step_fused()
        p1 = mc.perturb(state["p"])
        mis1, chi1, L1 = mc.misfit(p1)
        better = chi1 < state["chi"]
        u = mc.proposer.uniform(C)
        acc = better | (~better & (u > 1.0 - torch.exp(-(chi1 - state["chi"]) / 2.0)))
        state["p"] = torch.where(acc[:, None], p1, state["p"])
        state["chi"] = torch.where(acc, chi1, state["chi"])
        state["acc"] = acc

    setup()
    n0 = mc.n_forward
    from pysurfinv_amd import _lib, forward
    ring = forward.EventRing(K)                            # HIP events on the launch stream around the solver's kernels
    emc = cg.children[0] if cg is not None else mc         # (chain groups: the first group's stream)
    emc.event_ring = ring
    elapsed = rt.timed(step, K, W)
    kms = ring.kernel_ms().mean(axis=0)
    emc.event_ring = None
    if cg is not None:
        cg.join()
    L = int(mb.to_model(state["p"][:4])[0].shape[2])
    acc_rate, = rt.max_over_ranks(float((track_row[:, 2] if fused else state["acc"].double()).mean()))
    per_rank_ms, = rt.max_over_ranks(elapsed / K * 1e3)
    Cl = (cg.bounds[1] - cg.bounds[0]) if cg is not None else C          # chains per launch (one chain group)
    team = int(_lib.lib().surfdisp_get_team2(Cl, L, len(MCMC_PERIODS), _lib.KIND_RAYLEIGH | _lib.PHASE_ONLY))
    return {"metric": "Metropolis steps/s, model3D grid share (BASELINE configs[3])", "unit": "steps/s",
            "ms_per_step_max_over_ranks": per_rank_ms,
            "kernel_ms": {"prep": kms[0], "phase": kms[1], "finish": kms[2],
                          "how": "HIP events on the launch stream around the solver's kernels of the K timed lock steps"
                                 + (f" (first of {cg.G} chain groups, {Cl} chains per launch; the groups' streams share the chip, so a "
                                    "kernel's duration includes the time it shares SIMDs with the other group's kernels)" if cg is not None else "")},
            "roofline": leg_roofline("grid", kms[1], (20 * L + 4 * len(MCMC_PERIODS)) * Cl, team=team,
                                     chip=(cg.G, elapsed / K * 1e3) if cg is not None else None),
            "value": (total * chains if total else rt.world * C) * K / elapsed,
            "forward_solves_per_s": (total * chains if total else rt.world * C) * K / elapsed,
            "ms_per_step": elapsed / K * 1e3, "steps": K, "warmup": W, "n_gpus": rt.world, "scaling": "strong" if total else "weak",
            "config": {"workload": (f"BASELINE configs[3], the WHOLE grid on every rank count: {total} points x {chains} chains dealt "
                                    f"to {rt.world} rank(s) in contiguous blocks, " if total else
                                    "BASELINE configs[3] share per GPU: 512 points x 50 chains, ") +
                                   "96-layer continental model, 19 periods, Rayleigh phase-only misfit, default scan",
                       "total_points": total if total else rt.world * pts,
                       "lock_step": ("fused: propose kernel, parameters->stacks, prep / root search / finish, accept kernel"
                                     if fused else "torch glue around the solver"),
                       "chain_groups": cg.G if cg is not None else 1,
                       "prior_rules": "PriorRules(sediment + crust monotone, positive jumps, Vs <= 4.9): masked redraw rounds on the device"
                                      if mc.isgood is not None else None,
                       "points_per_gpu": pts, "chains_per_point": chains, "chains_per_gpu": C, "layers": L,
                       "periods": len(MCMC_PERIODS)},
            "accept_rate_last_step": acc_rate, "forward_solves_timed_this_rank": int(mc.n_forward - n0)}


def workload_mcmc(rt, args, steps=None, warmup=None):
    """BASELINE configs[2]: one surface point, 100 000 Metropolis steps as 100 chains x 1000 (the reference's
    MCinvMP layout, point.py:90-125), here the rate of the 100 chains' steps: the library's default (speculative lock
    steps), one step per solve, the (stack, period) decomposition, and the torch-glue step replayed from a HIP graph."""
    from pysurfinv_amd.mcmc import MetropolisBatch
    K = steps if steps is not None else max(args.steps, 50)
    mb, c_obs, unc = _mcmc_setup(rt, 1, 100)
    out = {"metric": "Metropolis steps/s, single point (BASELINE configs[2])", "unit": "steps/s", "n_gpus": rt.world,
           "config": {"workload": "BASELINE configs[2]: one point, 100 chains in lock step (100 000 steps = 1000 lock steps), "
                                  "96-layer continental model, 19 periods, Rayleigh phase-only misfit", "chains": 100}}
    mc = MetropolisBatch(mb.spec, mb.to_model, MCMC_PERIODS, c_obs[0], unc[0], device=rt.dev, seed=3 + rt.rank)
    from pysurfinv_amd import _lib, forward
    # the sampler as the library runs it for 100 chains: speculative lock steps (MetropolisBatch.auto_spec_depth = 4: the
    # tree of the next four accept / reject outcomes, 15 proposals per chain, ONE batched solve of 1 500 stacks, four
    # Metropolis steps; the chain is distributed as the plain sampler's) - and, beside it, one step per solve
    d = mc.auto_spec_depth(100)
    M = (1 << d) - 1
    K3 = -(-K // d) * d                                    # timed steps: whole lock steps
    mc.run(100, 1 + 2 * d); rt.barrier()
    mc.event_ring = forward.EventRing(K3 // d + 1)
    t0 = time.perf_counter(); mc.run(100, K3 + 1); rt.barrier()
    dt, = rt.max_over_ranks(time.perf_counter() - t0)
    kms = mc.event_ring.kernel_ms()[1:].mean(axis=0)        # (slot 0: the start models' own solve)
    mc.event_ring = None
    L = int(mb.to_model(mc.reset(2))[0].shape[2])
    out.update({"value": rt.world * 100 * K3 / dt, "ms_per_step": dt / K3 * 1e3, "ms_per_lock_step": dt / (K3 // d) * 1e3,
                "spec_depth": d, "stacks_per_lock_step": 100 * M, "steps": K3, "scaling": "weak",
                "kernel_ms": {"prep": kms[0], "phase": kms[1], "finish": kms[2]},
                "roofline": leg_roofline("mcmc", kms[1], (20 * L + 4 * len(MCMC_PERIODS)) * 100 * M,
                                         team=int(_lib.lib().surfdisp_get_team2(100 * M, L, len(MCMC_PERIODS), _lib.KIND_RAYLEIGH | _lib.PHASE_ONLY)))})
    if os.environ.get("BENCH_MCMC_ONLY_DEFAULT") == "1":   # counter passes: one launch size per kernel instantiation
        return out
    mc.run(100, 4, spec_depth=1); rt.barrier()
    t0 = time.perf_counter(); mc.run(100, K + 1, spec_depth=1); rt.barrier()
    dt1, = rt.max_over_ranks(time.perf_counter() - t0)
    out.update({"value_one_step_per_solve": rt.world * 100 * K / dt1, "ms_per_lock_step_one_step_per_solve": dt1 / K * 1e3})
    # the same lock step with the opt-in (stack, period) decomposition for small chain counts (independent="auto")
    mca = MetropolisBatch(mb.spec, mb.to_model, MCMC_PERIODS, c_obs[0], unc[0], device=rt.dev, seed=3 + rt.rank, independent="auto")
    mca.run(100, 4); rt.barrier()
    t0 = time.perf_counter(); mca.run(100, K + 1); rt.barrier()
    dta, = rt.max_over_ranks(time.perf_counter() - t0)
    out.update({"value_independent_auto": rt.world * 100 * K / dta, "ms_per_lock_step_independent_auto": dta / K * 1e3,
                "independent_auto_note": "opt-in: MetropolisBatch(independent='auto') / Point.MCinvMP(independent='auto'); "
                                         "equal to the faithful walk to 4e-6 on this model (scripts/indep_vs_faithful.py)"})
    try:
        if os.environ.get("BENCH_MCMC_NO_GRAPH") == "1":   # counter passes: rocprofv3 --pmc and HIP-graph replay do not mix
            raise RuntimeError("skipped (BENCH_MCMC_NO_GRAPH=1)")
        mc.run_graphed(100, 8); rt.barrier()
        t0 = time.perf_counter(); mc.run_graphed(100, 4 * K + 2); rt.barrier()
        dtg, = rt.max_over_ranks(time.perf_counter() - t0)
        out.update({"value_hip_graph": rt.world * 100 * 4 * K / dtg, "ms_per_lock_step_hip_graph": dtg / (4 * K) * 1e3,
                    "seconds_for_100000_steps_hip_graph": 1000 * dtg / (4 * K)})
    except Exception as e:
        out["hip_graph_error"] = repr(e)[:200]
    return out


def workload_c5(rt, args, steps=None, warmup=None):
    """BASELINE configs[4] per GPU: 16 384 ThermSeis-derived 64-layer stacks (water + Cascadia sediment + crust + thermal
    mantle: parameters -> stacks on the device), joint Rayleigh + Love phase + group velocity at 20 periods, and the
    analytic sensitivity kernels dc/d(Vs, Vp, rho) of every layer."""
    torch = rt.torch
    from pysurfinv_amd import _lib, forward, senskernel, synth
    from pysurfinv_amd.brownian import TorchProposer
    from pysurfinv_amd.layers_batch import Model1DBatch
    B = int(os.environ.get("BENCH_C5_STACKS", "16384"))
    K = steps if steps is not None else max(3, args.steps // 4)
    W = warmup if warmup is not None else 1
    per = torch.from_numpy(synth.default_periods(20)).to(rt.dev)
    mb = Model1DBatch(C5_SETTING, device=rt.dev)
    params = TorchProposer(mb.spec, rt.dev, seed=1 + rt.rank).reset(B)
    st = {}

    def gen():
        st["model"], st["nlay"] = mb.to_model(params)
    t_gen = rt.timed(gen, K, W)
    L = int(st["model"].shape[2])
    plan = forward.JointPlan(B, L, 20, device=rt.dev)
    ringR, ringL = forward.EventRing(K), forward.EventRing(K)
    cnt = {"i": 0}

    def joint():
        plan.run(st["model"], per, nlay=st["nlay"], events=(ringR.slot(cnt["i"]), ringL.slot(cnt["i"])))
        cnt["i"] += 1
    t_joint = rt.timed(joint, K, W)
    kmsR, kmsL = ringR.kernel_ms().mean(axis=0), ringL.kernel_ms().mean(axis=0)
    t_kr = rt.timed(lambda: senskernel.analytic_kernels(st["model"], per, wtype="R", nlay=st["nlay"]), K, W)
    t_kl = rt.timed(lambda: senskernel.analytic_kernels(st["model"], per, wtype="L", nlay=st["nlay"]), K, W)
    out = plan.run(st["model"], per, nlay=st["nlay"]); rt.barrier()
    okR, okL = float((out["statusR"] == 0).float().mean()), float((out["statusL"] == 0).float().mean())
    w = rt.world
    roofR = leg_roofline("c5", kmsR[1], (20 * L + 8 * 20) * B,
                         team=int(_lib.lib().surfdisp_get_team2(B, L, 20, _lib.KIND_RAYLEIGH | _lib.PIPELINED)))
    roofL = leg_roofline("c5", kmsL[1], (20 * L + 8 * 20) * B, kernel_prefix="surfdisp_phase_kernel<1,",
                         team=int(_lib.lib().surfdisp_get_team2(B, L, 20, _lib.KIND_LOVE | _lib.PIPELINED)))
    roofR["chip_level"] = chip_level_share("c5", [roofR.get("kernel"), roofL.get("kernel"), "surfdisp_ellip_kernel",
                                                  "surfdisp_group_kernel<2,0>", "surfdisp_group_kernel<1,0>"], t_joint / K * 1e3)
    return {"metric": "joint R+L c+U forward solves/s + sensitivity kernels, 64-layer thermal stacks (BASELINE configs[4])",
            "unit": "solves/s", "value": w * 2 * B * K / t_joint, "n_gpus": w, "steps": K, "scaling": "weak",
            "config": {"workload": "BASELINE configs[4] per GPU: 16384 ThermSeis-derived stacks, joint Rayleigh+Love c+U at 20 "
                                   "periods, analytic sensitivity kernels", "stacks_per_gpu": B, "layers": L, "periods": 20},
            "ms_thermal_parameters_to_stacks": t_gen / K * 1e3, "ms_joint_R_L_c_U": t_joint / K * 1e3,
            "ms_forward_plus_kernels_R": t_kr / K * 1e3, "ms_forward_plus_kernels_L": t_kl / K * 1e3,
            "kernel_sets_per_s_R": w * B * K / t_kr, "kernel_sets_per_s_L": w * B * K / t_kl,
            "solved_fraction_R": okR, "solved_fraction_L": okL,
            "kernel_ms": {"rayleigh": {"prep": kmsR[0], "phase": kmsR[1], "group_and_finish": kmsR[2]},
                          "love": {"prep": kmsL[0], "phase": kmsL[1], "group_and_finish": kmsL[2]},
                          "how": "HIP events on each plan's own stream; the two streams share the chip, so a kernel's "
                                 "duration includes the time it shares SIMDs with the other wave type's kernels"},
            "roofline": roofR, "roofline_love_root_search": roofL}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["all", "forward", "grid", "mcmc", "c5", "launchcheck"], default="all",
                    help="all (default): the forward headline line + short legs of the other BASELINE configs inside it; "
                         "a single name: only that workload, as the line itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--total-points", type=int, default=0,
                    help="--workload grid only: STRONG scaling - the SAME grid of this many surface points (4096 = BASELINE "
                         "configs[3]) x 50 chains is dealt to the ranks in contiguous blocks (1 GPU: all of it, 8 GPUs: 512 "
                         "points each); every point has the same observations and every chain the same random stream whatever "
                         "the rank count.  0 (default): weak scaling, 512 points per rank")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: this process stays a plain launcher (no torch, no HIP); see launch_ranks
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    from pysurfinv_amd import _lib
    rt = Runtime(args)
    if args.workload == "launchcheck":
        # what every rank was handed by the launcher, gathered over the process group: proves the N > 1 entry without
        # a solver call (tests/test_bench_launch.py runs it on CPU with BENCH_REHEARSAL=cpu)
        mine = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        mine["pid"], mine["ppid"] = os.getpid(), os.getppid()
        if args.total_points:                                  # the block of the strong-scaling grid this rank would own
            from pysurfinv_amd.shard import shard_range
            mine["points"] = list(shard_range(args.total_points, rt.rank, rt.world))
        ranks = [mine]
        if rt.dist is not None:
            ranks = [None] * rt.world
            rt.dist.all_gather_object(ranks, mine)
        t, = rt.max_over_ranks(0.001 * (rt.rank + 1))
        if rt.rank == 0:
            print(json.dumps(rt.stamp({"metric": "launchcheck", "n_gpus": rt.world, "ranks": ranks, "max_over_ranks_check": t})), flush=True)
        if rt.dist is not None:
            rt.dist.barrier(); rt.dist.destroy_process_group()
        return
    if _lib.lib().surfdisp_device_count() < 1:
        raise SystemExit("no HIP device: the product path has no CPU fallback")

    single = {"grid": workload_grid, "mcmc": workload_mcmc, "c5": workload_c5}
    if args.workload in single:
        line = single[args.workload](rt, args)
        line.update({"higher_is_better": True, "data": "synthetic", "vs_baseline": None})
    else:
        line = workload_forward(rt, args)
        if args.workload == "all":
            extra = {}
            legs = {"grid": (workload_grid, dict(steps=10, warmup=2)), "mcmc": (workload_mcmc, dict(steps=60)),
                    "c5": (workload_c5, dict(steps=3, warmup=1))}
            for name in os.environ.get("BENCH_LEGS", "mcmc,grid,c5").split(","):       # development: subset / order
                fn, kw = legs[name]
                try:
                    extra[name] = fn(rt, args, **kw)
                except Exception as e:                          # a side leg must never cost the headline line
                    extra[name] = {"error": repr(e)[:300]}
                    rt.barrier()
            if line is not None:
                line["workloads"] = extra
    if rt.rank == 0 and line is not None:
        print(json.dumps(rt.stamp(line)), flush=True)
    if rt.dist is not None:
        rt.dist.barrier()
        rt.dist.destroy_process_group()


if __name__ == "__main__":
    main()
