#!/bin/bash
# Cache-behaviour PMC passes (vector L1 and L2) for one workload.  Usage: tools/gpu_pmc_cache.sh <tag> [workload] [mode]
set -o pipefail
TAG=${1:-r01}; WL=${2:-bunny}; MODE=${3:-fast}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 tools/profile_run.py $WL 3 $MODE > $OUT/pmc_$name.txt 2>&1 || { echo "pass $name failed"; tail -3 $OUT/pmc_$name.txt; return 1; }
}
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum &&
pass tcp2 TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum &&
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum &&
pass ta TA_BUSY_avr TA_TA_BUSY_sum
python3 tools/summarize_pmc.py $OUT | grep -A30 "k_trace" 
