#!/bin/bash
# GPU box: group-velocity kernel, workgroup order (SURFDISP_GROUP_ORDER = 0 plain period-major, g: an XCD takes g stack blocks at a
# time): durations and HBM fetch of scripts/time_kernels.py for the c5 shape (16 384 x L64) and a 25 600 x L96 batch.
# usage: [NOFETCH=1] scripts/group_order_probe.sh "0 1 2 4"   (+100: periods in descending order)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for o in ${1:-0 1}; do
  for shape in "16384 64" "25600 96"; do
    set -- $shape
    export SURFDISP_GROUP_ORDER=$o TK_B=$1 TK_L=$2
    d=$R/gpurun_out/prof_go_${o}_$2
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 $R/scripts/time_kernels.py > $d.trace.log 2>&1 || exit 1
    if [ -z "${NOFETCH:-}" ]; then
      timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 $R/scripts/time_kernels.py > $d.fetch.log 2>&1 || exit 1
    fi
    echo "=== order $o L$2"
    python3 $R/scripts/kstats.py $d/trace | grep "group"
    if [ -z "${NOFETCH:-}" ]; then python3 $R/scripts/pmc_sum.py $d/fetch FETCH_SIZE 2048 | grep "group"; fi
  done
done
