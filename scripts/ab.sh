#!/bin/bash
# GPU box: A/B of pysurfinv_amd/lib/libsurfdisp_hip.so (A) against libsurfdisp_var.so (B, `make -C pysurfinv_amd/csrc
# variant EXTRA=-D...`), alternating runs on the same box; prints value / one-in-flight / phase ms per run.
cd "$(dirname "$0")/.."
N=${1:-3}
for i in $(seq $N); do
  for v in A B; do
    if [ $v = B ]; then export SURFDISP_LIB_PATH=$PWD/pysurfinv_amd/lib/libsurfdisp_var.so; else unset SURFDISP_LIB_PATH; fi
    python bench.py --workload forward --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v  value %.2f M  one-in-flight %.2f M  phase %.4f ms  group %.4f ms' % (d['value']/1e6, d['value_one_batch_in_flight']/1e6, d['kernel_ms']['phase'], d['kernel_ms']['group_and_finish']))"
  done
done
