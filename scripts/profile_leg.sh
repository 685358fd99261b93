#!/bin/bash
# scripts/profile_leg.sh <tag> <leg> [extra bench args]  -- run on the GPU box (via gpurun).
# rocprofv3 passes of ONE bench leg (grid | c5 | mcmc | forward):
#   trace   : --kernel-trace --stats (average duration of every kernel instantiation)
#   pmc_*   : counter passes, each its own run with --kernel-trace only (gpurun's rule): HBM bytes, VALU issue,
#             lane utilisation, wait buckets, LDS activity / bank conflicts / LDS issue stalls, occupancy.
# Output under gpurun_out/prof_<tag>_<leg>/ ; scripts/summarise_leg.py condenses it into profiles/<tag>/.
set -uo pipefail
TAG=${1:-r03a}
LEG=${2:-grid}
shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${LEG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
case $LEG in
  grid) STEPS="--steps 6 --warmup 2" ;;
  c5)   STEPS="--steps 8 --warmup 1" ;;
  mcmc) STEPS="--steps 30 --warmup 2"; export BENCH_MCMC_NO_GRAPH=1 BENCH_MCMC_ONLY_DEFAULT=1 ;;   # (a --pmc pass hung on the HIP-graph replay of this leg; only the default sampler: one launch size per kernel)
  *)    STEPS="--steps 5 --warmup 1 --no-cpu-baseline"; export BENCH_IN_FLIGHT=1 ;;
esac
BENCH="python3 $ROOT/bench.py --workload $LEG $STEPS $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
pass() {   # name counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
  local rc=$?
  echo "pmc $name rc=$rc"
  if [ $rc -ge 124 ]; then echo "pass $name timed out: stopping (no further GPU step after a killed one)"; exit 1; fi
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq  SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq3 SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 GRBM_GUI_ACTIVE
pass sq4 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
grep -l "rror" "$OUT"/pmc_*.log 2>/dev/null | head
exit 0
