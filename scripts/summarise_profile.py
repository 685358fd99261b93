#!/usr/bin/env python3
"""scripts/summarise_profile.py <tag>: condense gpurun_out/prof_<tag>/ (written by scripts/profile.sh on
the GPU box) into the tracked summaries profiles/<tag>/{kernel_stats.csv, pmc_summary.csv} and
profiles/traffic_latest.json (read by bench.py for roofline.traffic / roofline.valu_pmc).

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB
(x1024) and FETCH_SIZE under-reports by 2x on gfx950 (checked on the prep kernel, whose read is the
13.1 MB model array)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _source_hash():
    sys.path.insert(0, ROOT)
    from pysurfinv_amd import _lib
    return _lib.source_hash()
KERNELS = ("surfdisp_prep_kernel", "surfdisp_phase_kernel", "surfdisp_group_kernel", "surfdisp_finish_kernel")
SIMDS = 1024


def short(name):
    """Kernel family of a trace/counter row.  Template arguments of the root search: <KIND, G, INDEP, FAST, EXACT>.
    Kept out of the summary: its opt-in fast-scan instantiation (FAST = true; bench.py times it beside the
    headline), the exact fallback kernel (EXACT = true: launched behind every root search, normally idle) and the
    group kernel with analytic partials (last template argument true)."""
    for k in KERNELS:
        if k in name:
            targs = name.split("<")[-1].split(">")[0].replace(" ", "").split(",")
            if k == "surfdisp_phase_kernel" and (targs[-1] == "true" or targs[-2] == "true"):
                return None
            if k == "surfdisp_group_kernel" and targs[-1] == "true":
                return None
            return k
    return None


COST_PLAIN, COST_TRANS, COST_F64 = 2.2, 8.1, 4.2      # SIMD cycles per wave-level instruction, measured


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
    stats = glob.glob(os.path.join(src, "trace_one", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "kernel_stats_one_in_flight.csv"))
    acc = defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                a = acc[(k, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    mean = {key: v[0] / v[1] for key, v in acc.items()}
    with open(os.path.join(dst, "pmc_summary.csv"), "w") as fh:
        fh.write("kernel,Counter_Name,mean,count\n")
        for (k, c), v in acc.items():
            fh.write(f"{k},{c},{v[0] / v[1]},{v[1]}\n")
    per_kernel, valu = {}, {}
    for k in KERNELS:
        if (k, "FETCH_SIZE") not in mean:
            continue
        fb = mean[(k, "FETCH_SIZE")] * 1024 * 2
        wb = mean[(k, "WRITE_SIZE")] * 1024
        per_kernel[k] = {"fetch_bytes": fb, "write_bytes": wb, "total": fb + wb}
        if k in ("surfdisp_phase_kernel", "surfdisp_group_kernel") and (k, "SQ_INSTS_VALU") in mean:
            cyc = mean[(k, "GRBM_GUI_ACTIVE")] / 8                    # summed over the 8 XCDs
            wc = mean[(k, "SQ_WAVE_CYCLES")]
            valu[k] = {
                "valu_wave_instructions": mean[(k, "SQ_INSTS_VALU")],
                "kernel_cycles": cyc,
                "valu_issue_frac_2cyc": mean[(k, "SQ_INSTS_VALU")] * 2 / (SIMDS * cyc),
                "lane_utilisation": mean[(k, "SQ_THREAD_CYCLES_VALU")] / (mean[(k, "SQ_ACTIVE_INST_VALU")] * 64),
                "wave_active_frac": mean[(k, "SQ_ACTIVE_INST_ANY")] / wc,
                "wave_wait_inst_frac": mean[(k, "SQ_WAIT_INST_ANY")] / wc,
                "wave_wait_any_frac": mean[(k, "SQ_WAIT_ANY")] / wc,
            }
            # the same share priced with the issue costs measured on this chip (scripts/microbench/valu_rates.hip,
            # profiles/r02e/valu_rates.txt, 4 wavefronts per SIMD): plain fp32 2.2 cycles, transcendental 8.1, fp64 4.2
            tr32 = mean.get((k, "SQ_INSTS_VALU_TRANS_F32"))
            if tr32 is not None:
                f64 = sum(mean.get((k, "SQ_INSTS_VALU_%s_F64" % t), 0.0) for t in ("ADD", "MUL", "FMA"))
                tr64 = mean.get((k, "SQ_INSTS_VALU_TRANS_F64"), 0.0)
                plain = mean[(k, "SQ_INSTS_VALU")] - tr32 - f64 - tr64
                valu[k]["valu_issue_frac_measured_costs"] = (plain * COST_PLAIN + (tr32 + tr64) * COST_TRANS + f64 * COST_F64) / (SIMDS * cyc)
                valu[k]["instruction_classes"] = {"plain": plain, "transcendental": tr32 + tr64, "fp64": f64}
    import hashlib
    with open(os.path.join(ROOT, "pysurfinv_amd", "lib", "libsurfdisp_hip.so"), "rb") as fh:
        lib_hash = hashlib.sha256(fh.read()).hexdigest()[:16]
    out = {
        "round": tag,
        "lib_sha256_16": lib_hash,          # bench.py compares it with the library it runs: a stale profile is visible
        "src_sha256_16": _source_hash(),    # ... and, since hipcc's output is not bit-reproducible, with the sources a rebuilt library came from
        "workload": "B=65536 L=10 P=20 Rayleigh c+U, default (point-by-point) scan, one batch in flight",
        "phase_kernel_hbm_bytes_per_launch": per_kernel.get("surfdisp_phase_kernel", {}).get("total"),
        "per_kernel": per_kernel,
        "valu": valu,
        "method": "rocprofv3 --pmc in separate passes (scripts/profile.sh, scripts/summarise_profile.py); "
                  "FETCH_SIZE/WRITE_SIZE KB->bytes x1024; FETCH_SIZE x2 (gfx950 correction, calibrated on the prep "
                  "kernel's 13.1 MB model read); VALU issue share = SQ_INSTS_VALU*2 cycles / (1024 SIMDs * "
                  "GRBM_GUI_ACTIVE/8); valu_issue_frac_measured_costs prices plain / transcendental / fp64 instructions "
                  "at 2.2 / 8.1 / 4.2 SIMD cycles (scripts/microbench/valu_rates.hip)",
    }
    with open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["valu"], indent=1))
    print({k: round(v["total"] / 1e6, 1) for k, v in per_kernel.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
