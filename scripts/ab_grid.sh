#!/bin/bash
# GPU box: A/B (see ab.sh) on the grid leg's lock step and the c5 leg
cd "$(dirname "$0")/.."
for i in 1 2; do for v in A B; do
  if [ $v = B ]; then export SURFDISP_LIB_PATH=$PWD/pysurfinv_amd/lib/libsurfdisp_var.so; else unset SURFDISP_LIB_PATH; fi
  python bench.py --workload grid --steps 10 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v grid %.3f ms per lock step  %.2f M steps/s' % (d['ms_per_step'], d['value']/1e6))"
  python bench.py --workload c5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c5 joint %.3f ms  kernels R %.3f ms L %.3f ms' % (d['ms_joint_R_L_c_U'], d['ms_forward_plus_kernels_R'], d['ms_forward_plus_kernels_L']))"
done; done
