#!/usr/bin/env python3
"""Print the surfdisp kernels of a rocprofv3 --kernel-trace --stats output directory (calls, average us)."""
import csv, glob, sys
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    print("==", d, "" if fs else "(no kernel_stats.csv)")
    for f in fs[:1]:
        for r in csv.DictReader(open(f)):
            if "surfdisp" in r["Name"]:
                print(f"   {r['Name'][:78]:78s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:9.1f} us")
