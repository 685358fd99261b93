#!/bin/bash
# GPU box: A (libsurfdisp_hip.so) against B (libsurfdisp_var.so) over lanes per stack x batches in flight:
# which instantiation of the root search a change helps or hurts.   scripts/ab_matrix.sh "2 4" "1 3"
cd "$(dirname "$0")/.."
for team in ${1:-2 4}; do for nf in ${2:-1 3}; do for v in A B A B; do
  if [ $v = B ]; then export SURFDISP_LIB_PATH=$PWD/pysurfinv_amd/lib/libsurfdisp_var.so; else unset SURFDISP_LIB_PATH; fi
  SURFDISP_TEAM=$team BENCH_IN_FLIGHT=$nf python bench.py --workload forward --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('team $team in-flight $nf $v  value %.2f M  one-in-flight %.2f M  phase %.4f ms' % (d['value']/1e6, d['value_one_batch_in_flight']/1e6, d['kernel_ms']['phase']))"
done; done; done
