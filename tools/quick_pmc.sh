#!/bin/bash
# quick instruction-count passes (subset of tools/gpu_profile.sh)
set -o pipefail
TAG=${1:-q}; WL=${2:-bunny}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SOLO="bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $WL --flights 1"
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $SOLO > $OUT/pmc_$name.txt 2>&1 || { echo "pass $name failed"; tail -3 $OUT/pmc_$name.txt; return 1; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pass sq3 SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM &&
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*_counter_collection.csv" -delete
grep -A40 "k_shadow_shade_packet\|k_trace_packet<0" $OUT/summary.txt | grep -v "^--" | head -90
