#!/bin/bash
# GPU box: one library, several environment settings: bench legs under each "NAME:VAR=val,VAR=val" argument.
# usage: scripts/ab_env.sh "legs" cfg1 cfg2 ...   (legs: any of fwd grid c5 mcmc, comma separated)
cd "$(dirname "$0")/.."
LEGS=$1; shift
for cfg in "$@"; do
  name=${cfg%%:*}; vars=${cfg#*:}
  envs=$(echo "$vars" | tr ',' ' ')
  for leg in $(echo $LEGS | tr ',' ' '); do
    case $leg in
      fwd)  env $envs python bench.py --workload forward --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name fwd  value %.2f M  one-in-flight %.2f M  phase %.4f ms  group %.4f ms  team %s/%s' % (d['value']/1e6, d['value_one_batch_in_flight']/1e6, d['kernel_ms']['phase'], d['kernel_ms']['group_and_finish'], d['config']['team_lanes'], d['config']['team_lanes_one_batch_in_flight']))" ;;
      grid) env $envs python bench.py --workload grid --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name grid %.3f ms per lock step  phase %.3f ms  %.2f M steps/s' % (d['ms_per_step'], d['kernel_ms']['phase'], d['value']/1e6))" ;;
      c5)   env $envs python bench.py --workload c5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$name c5 joint %.3f ms  kernels R %.3f ms L %.3f ms | R phase %.3f group %.3f  L phase %.3f group %.3f' % (d['ms_joint_R_L_c_U'], d['ms_forward_plus_kernels_R'], d['ms_forward_plus_kernels_L'], k['rayleigh']['phase'], k['rayleigh']['group_and_finish'], k['love']['phase'], k['love']['group_and_finish']))" ;;
      mcmc) env $envs python bench.py --workload mcmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name mcmc %.4f ms per lock step  auto %.4f ms  graph %.4f ms' % (d['ms_per_lock_step'], d['ms_per_lock_step_independent_auto'], d.get('ms_per_lock_step_hip_graph', float('nan'))))" ;;
    esac
  done
done
