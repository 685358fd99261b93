#!/bin/bash
# GPU box: the grid leg's lock step (25 600 chains x 96 layers, phase only) for lanes per stack x priority balance
cd "$(dirname "$0")/.."
for team in 8 16 32 64; do for bal in 0 1; do
  r=$(SURFDISP_TEAM=$team SURFDISP_BALANCE=$bal python bench.py --workload grid --steps 8 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms per lock step, %.2f M steps/s' % (d['ms_per_step'], d['value']/1e6))")
  echo "team=$team balance=$bal : $r"
done; done
