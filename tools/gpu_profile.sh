#!/bin/bash
# Runs on the GPU box: kernel traces + the PMC passes over the benchmark command itself.
# Usage: tools/gpu_profile.sh <tag>_<workload> [workload] [mode]      (e.g. r03_bunny bunny)
#   trace       rocprofv3 --kernel-trace --stats of the default command (frames in flight: kernels overlap)
#   trace_solo  the same with --flights 1: every kernel alone on the GPU; these are the durations the
#               bench line's roofline is computed from, and the counter passes run the same way
#   pmc_*       one rocprofv3 run per counter group (counters are not split over passes automatically)
set -o pipefail
TAG=${1:-r02}; WL=${2:-bunny}; MODE=${3:-fast}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --workload $WL --mode $MODE"
SOLO="$BENCH --flights 1"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.txt 2>&1 || { tail -5 $OUT/trace.txt; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_solo -- python3 $SOLO > $OUT/trace_solo.txt 2>&1 || { tail -5 $OUT/trace_solo.txt; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $SOLO --steps 3 --warmup 1 > $OUT/pmc_$name.txt 2>&1 || { echo "pass $name failed"; tail -3 $OUT/pmc_$name.txt; return 1; }
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pass sq3 SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM &&
pass sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_REQ SQC_TC_STALL &&
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum &&
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 tools/summarize_pmc.py $OUT --json $OUT/summary.json --traffic-json $OUT/pmc_traffic.json --workload $WL > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
# the bench line last, with this run's counts in place: its issue-rate and traffic figures are computed
# from profiles/pmc_traffic.json and must belong to the kernels that were just measured
python3 - "$OUT/pmc_traffic.json" <<'PY'
import json, sys
new = json.load(open(sys.argv[1]))
try:
    cur = json.load(open("profiles/pmc_traffic.json"))
except Exception:
    cur = {}
cur.update(new)
json.dump(cur, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
cp profiles/pmc_traffic.json $OUT/pmc_traffic_merged.json
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --workload $WL --mode $MODE > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
# keep only the small files: the raw per-dispatch CSVs are large
find $OUT -name "*_counter_collection.csv" -size +8M -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
