# development: bench.py over team size x batches in flight (run on the GPU box)
for t in ${TEAMS:-1 2 4}; do for f in ${FLIGHTS:-2 3 4 6}; do export SURFDISP_TEAM=$t BENCH_IN_FLIGHT=$f; python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('team',d['config']['team_lanes'],'inflight',d['config']['batches_in_flight'],round(d['value']/1e6,2), round(d['value_one_batch_in_flight']/1e6,2), round(d['kernel_ms']['phase'],3))
"; done; done
