#!/bin/bash
# Runs on the GPU box: kernel trace + the PMC passes for one workload.
# Usage: tools/gpu_profile.sh <tag> [workload] [mode]
set -o pipefail
TAG=${1:-r01}; WL=${2:-bunny}; MODE=${3:-fast}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python3 tools/profile_run.py $WL 5 $MODE > $OUT/plain.txt 2>&1 || { tail -5 $OUT/plain.txt; exit 1; }
cat $OUT/plain.txt
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/profile_run.py $WL 5 $MODE > $OUT/trace.txt 2>&1 || { tail -5 $OUT/trace.txt; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 tools/profile_run.py $WL 3 $MODE > $OUT/pmc_$name.txt 2>&1 || { echo "pass $name failed"; tail -3 $OUT/pmc_$name.txt; return 1; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -name "*.csv" | head -30
