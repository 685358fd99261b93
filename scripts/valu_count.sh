#!/bin/bash
# GPU box: VALU wave-instructions per launch of the root-search kernels of the forward leg (one batch in flight) for several library
# builds: scripts/valu_count.sh name:path ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  name=${cfg%%:*}; lib=${cfg#*:}
  d=$R/gpurun_out/valu_$name
  SURFDISP_LIB_PATH=$lib BENCH_IN_FLIGHT=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $d -- python3 $R/bench.py --workload forward --steps 5 --warmup 1 --no-cpu-baseline > $d.log 2>&1 || exit 1
  echo "== $name"
  python3 $R/scripts/pmc_sum.py $d SQ_INSTS_VALU 1 | grep "phase_kernel<2, 4\|group"
  python3 $R/scripts/pmc_sum.py $d SQ_INSTS_SALU 1 | grep "phase_kernel<2, 4"
done
