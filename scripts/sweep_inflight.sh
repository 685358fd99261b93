#!/bin/bash
# GPU box: bench headline for batches in flight x lanes per stack x launch hint x priority balance
cd "$(dirname "$0")/.."
for nf in 1 2 3; do for team in 0 2 4; do for hint in 1 0; do for bal in -1 0 1; do
  [ $nf = 1 ] && [ $hint = 0 ] && continue
  r=$(BENCH_IN_FLIGHT=$nf SURFDISP_TEAM=$team BENCH_PIPELINED_HINT=$hint SURFDISP_BALANCE=$bal python bench.py --workload forward --no-cpu-baseline --steps 20 --warmup 3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f M/s  (one in flight %.2f, fast scan %.2f)' % (d['value']/1e6, d['value_one_batch_in_flight']/1e6, d['value_fast_scan']/1e6))")
  echo "in_flight=$nf team=$team hint=$hint balance=$bal : $r"
done; done; done; done
