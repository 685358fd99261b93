#!/usr/bin/env python3
"""scripts/summarise_leg.py <tag> <leg>: condense gpurun_out/prof_<tag>_<leg>/ (scripts/profile_leg.sh on the GPU box)
into tracked summaries: profiles/<tag>/<leg>_kernel_stats.csv (rocprofv3 --stats, surfdisp kernels first),
profiles/<tag>/<leg>_pmc_summary.csv (mean of every counter per kernel INSTANTIATION) and profiles/traffic_<leg>.json
(read by bench.py for that leg's roofline block; tagged with the library hash).

Counter handling as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KB (x1024), FETCH_SIZE
x2 on gfx950; SQ_* cycle counters are quad-cycles summed over the SIMDs, GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 1024
COST_PLAIN, COST_TRANS, COST_F64 = 2.2, 8.1, 4.2      # SIMD cycles per wave-level instruction (profiles/r02e/valu_rates.txt)


def inst(name):
    """'void sd::surfdisp_phase_kernel<2, 16, false, false, false>(sd::PhaseArgs)' -> 'surfdisp_phase_kernel<2,16,0,0,0>'"""
    m = re.search(r"(surfdisp_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    t = (m.group(2) or "").replace(" ", "").replace("false", "0").replace("true", "1")
    return m.group(1) + t


def main(tag, leg):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{leg}")
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    dur = {}
    for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(dst, f"{leg}_kernel_stats.csv"), "w") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                k = inst(r["Name"])
                nm = k if k else r["Name"][:90]
                w.writerow([nm, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
                if k:
                    dur[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct_of_gpu_time": float(r["Percentage"])}
    acc = defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = inst(row["Kernel_Name"])
                if k is None:
                    continue
                a = acc[(k, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    mean = {key: v[0] / v[1] for key, v in acc.items()}
    with open(os.path.join(dst, f"{leg}_pmc_summary.csv"), "w") as fh:
        fh.write("kernel,Counter_Name,mean,count\n")
        for (k, c), v in sorted(acc.items()):
            fh.write(f"{k},{c},{v[0] / v[1]},{v[1]}\n")
    kernels = sorted({k for k, _ in mean})
    out_k = {}
    for k in kernels:
        g = lambda c, d=None: mean.get((k, c), d)
        e = dict(dur.get(k, {}))
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            e["hbm_bytes_per_launch"] = g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024
        if g("SQ_INSTS_VALU") is not None and g("GRBM_GUI_ACTIVE"):
            cyc = g("GRBM_GUI_ACTIVE") / 8
            wc = g("SQ_WAVE_CYCLES")
            e.update({"valu_wave_instructions": g("SQ_INSTS_VALU"), "kernel_cycles": cyc,
                      "valu_issue_frac_2cyc": g("SQ_INSTS_VALU") * 2 / (SIMDS * cyc),
                      "waves": g("SQ_WAVES"), "salu_per_valu": g("SQ_INSTS_SALU", 0) / g("SQ_INSTS_VALU"),
                      "lds_per_valu": g("SQ_INSTS_LDS", 0) / g("SQ_INSTS_VALU")})
            if wc:
                # SQ_WAVE_CYCLES: quad-cycles a wavefront is resident, summed over waves: / (cycles/4 * SIMDs) = waves per SIMD
                e["mean_waves_per_simd"] = wc / (cyc / 4 * SIMDS)
                for nm, c in (("wave_active_frac", "SQ_ACTIVE_INST_ANY"), ("wave_wait_inst_frac", "SQ_WAIT_INST_ANY"),
                              ("wave_wait_any_frac", "SQ_WAIT_ANY"), ("wave_wait_inst_lds_frac", "SQ_WAIT_INST_LDS"),
                              ("wave_active_valu_frac", "SQ_ACTIVE_INST_VALU"), ("wave_active_lds_frac", "SQ_ACTIVE_INST_LDS"),
                              ("wave_active_sca_frac", "SQ_ACTIVE_INST_SCA"), ("wave_active_vmem_frac", "SQ_ACTIVE_INST_VMEM")):
                    if g(c) is not None:
                        e[nm] = g(c) / wc
            if g("SQ_THREAD_CYCLES_VALU") is not None and g("SQ_ACTIVE_INST_VALU"):
                e["lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
            if g("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT", 0) / g("SQ_LDS_IDX_ACTIVE")
                e["lds_array_busy_frac_of_kernel"] = g("SQ_LDS_IDX_ACTIVE") / (cyc * 256)      # per CU
            tr32 = g("SQ_INSTS_VALU_TRANS_F32")
            if tr32 is not None:
                f64 = sum(g("SQ_INSTS_VALU_%s_F64" % t, 0.0) for t in ("ADD", "MUL", "FMA"))
                tr64 = g("SQ_INSTS_VALU_TRANS_F64", 0.0)
                plain = g("SQ_INSTS_VALU") - tr32 - f64 - tr64
                e["valu_issue_frac_measured_costs"] = (plain * COST_PLAIN + (tr32 + tr64) * COST_TRANS + f64 * COST_F64) / (SIMDS * cyc)
                e["instruction_classes"] = {"plain": plain, "transcendental": tr32 + tr64, "fp64": f64}
            for c in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_LEVEL_WAVES"):
                if g(c) is not None:
                    e[c.lower()] = g(c)
        out_k[k] = e
    with open(os.path.join(ROOT, "pysurfinv_amd", "lib", "libsurfdisp_hip.so"), "rb") as fh:
        lib_hash = hashlib.sha256(fh.read()).hexdigest()[:16]
    sys.path.insert(0, ROOT)
    from pysurfinv_amd import _lib
    out = {"round": tag, "leg": leg, "lib_sha256_16": lib_hash, "src_sha256_16": _lib.source_hash(), "kernels": out_k,
           "method": "rocprofv3 --kernel-trace --stats for durations; --pmc in separate passes (scripts/profile_leg.sh); "
                     "FETCH_SIZE/WRITE_SIZE KB->bytes x1024, FETCH_SIZE x2 (gfx950); VALU issue share = SQ_INSTS_VALU x 2 "
                     "cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8); *_frac = counter / SQ_WAVE_CYCLES"}
    with open(os.path.join(ROOT, "profiles", f"traffic_{leg}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k, e in out_k.items():
        keys = ("avg_us", "valu_issue_frac_2cyc", "valu_issue_frac_measured_costs", "lane_utilisation", "mean_waves_per_simd",
                "wave_active_frac", "wave_wait_inst_frac", "wave_wait_any_frac", "wave_wait_inst_lds_frac",
                "lds_bank_conflict_frac", "lds_array_busy_frac_of_kernel", "lds_per_valu", "salu_per_valu", "hbm_bytes_per_launch")
        print(k, {q: (round(e[q], 4) if isinstance(e.get(q), float) else e.get(q)) for q in keys if q in e})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
