#!/bin/bash
cd "$(dirname "$0")/.."
for nf in 3 4 5; do for team in 0 1 2 4; do
  r=$(BENCH_IN_FLIGHT=$nf SURFDISP_TEAM=$team python bench.py --workload forward --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f M/s  (fast scan %.2f)' % (d['value']/1e6, d['value_fast_scan']/1e6))")
  echo "in_flight=$nf team=$team : $r"
done; done
