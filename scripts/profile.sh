#!/bin/bash
# scripts/profile.sh <tag>  -- run on the GPU box (via gpurun).  Collects
#   (1) rocprofv3 --kernel-trace --stats of the default bench workload,
#   (2) PMC passes (separate runs, kernel-trace only as gpurun requires): HBM bytes, SQ activity.
# Output under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -uo pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload forward --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
# same trace with ONE batch in flight (what bench.py brackets with HIP events): kernel durations are
# not stretched by the second stream's kernels sharing the machine
export BENCH_IN_FLIGHT=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_one" -- python3 $ROOT/bench.py --workload forward --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/trace_one.log" 2>&1
echo "trace_one rc=$?"
# the counter passes keep one batch in flight too: the counters then describe the launch configuration
# whose kernel durations bench.py brackets with HIP events (with a second batch in flight the launches
# carry SURFDISP_PIPELINED and use two-lane teams)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "pmc write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
echo "pmc sq rc=$?"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
echo "pmc sq2 rc=$?"
find "$OUT" -name "*.csv" | head -40
unset BENCH_IN_FLIGHT
# instruction mix (fp64 runs at half rate: what bounds the group-velocity kernel)
export BENCH_IN_FLIGHT=1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$OUT/pmc_mix64" -- $BENCH > "$OUT/pmc_mix64.log" 2>&1
echo "pmc mix64 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT --output-format csv -d "$OUT/pmc_mix32" -- $BENCH > "$OUT/pmc_mix32.log" 2>&1
echo "pmc mix32 rc=$?"
unset BENCH_IN_FLIGHT
