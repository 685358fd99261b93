#!/usr/bin/env python3
"""Per-kernel average of one counter of a rocprofv3 --pmc output directory: pmc_sum.py <dir> <COUNTER> [scale]
(FETCH_SIZE / WRITE_SIZE are in KB: scale 1024; FETCH_SIZE additionally x 2 on gfx950, MI355X_MICROARCH.md)."""
import csv, glob, re, sys, collections
d, ctr = sys.argv[1], sys.argv[2]
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot, cnt = collections.Counter(), collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr or "surfdisp" not in r["Kernel_Name"]:
            continue
        seen[(r["Dispatch_Id"], re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void sd::", "").replace("sd::", ""))] += float(r["Counter_Value"])
    for (_, n), v in seen.items():
        tot[n] += v; cnt[n] += 1
for n in sorted(tot):
    print(f"   {n:60s} launches {cnt[n]:4d}  {ctr} per launch {tot[n] / cnt[n] * scale / 1e6:10.2f} M")
