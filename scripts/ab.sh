#!/bin/bash
# GPU box: A/B of pysurfinv_amd/lib/libsurfdisp_hip.so (A) against libsurfdisp_var.so (B: `make -C pysurfinv_amd/csrc
# variant EXTRA=-D...`, or a copy of an earlier build), alternating runs on the same box; prints value / one-in-flight /
# phase ms per run.  usage: scripts/ab.sh [rounds] [legs: fwd,grid,c5]
cd "$(dirname "$0")/.."
N=${1:-3}; LEGS=${2:-fwd}
for i in $(seq $N); do
  scripts/ab_env.sh $LEGS "A:SURFDISP_BALANCE=-1" "B:SURFDISP_LIB_PATH=$PWD/pysurfinv_amd/lib/libsurfdisp_var.so"
done
