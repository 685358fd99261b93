#!/bin/bash
# GPU box: VALU wave-instructions per launch of the Love root-search kernels of scripts/time_love.py for several library builds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  name=${cfg%%:*}; lib=${cfg#*:}
  d=$R/gpurun_out/valuL_$name
  SURFDISP_LIB_PATH=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $d -- python3 $R/scripts/time_love.py > $d.log 2>&1 || exit 1
  echo "== $name"
  python3 - $d <<'PY'
import csv, glob, sys, re, collections
tot=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="SQ_INSTS_VALU" and "phase_kernel<1" in r["Kernel_Name"] and "true>" not in r["Kernel_Name"].split("(")[0][-6:]:
            n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void sd::","")
            tot[(n, r["Grid_Size"])].append(float(r["Counter_Value"]))
for k,v in sorted(tot.items()): print("   %-55s grid %-9s launches %3d  VALU per launch %9.2f M" % (k[0], k[1], len(v), sum(v)/len(v)/1e6))
PY
done
