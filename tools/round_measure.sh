#!/bin/bash
# Runs on the GPU box: everything the round's tables are made of, one log each under gpurun_out/<tag>/.
# Usage: tools/round_measure.sh <tag> [profile|bench|parts|parity ...]   (default: all)
set -o pipefail
TAG=${1:-r03}; shift
WHAT=${*:-profile bench parts parity}
mkdir -p gpurun_out/$TAG
for w in $WHAT; do
  case $w in
    profile)
      for wl in bunny cornell_room spheres_1K; do
        tools/gpu_profile.sh ${TAG}_$wl $wl fast > gpurun_out/$TAG/profile_$wl.log 2>&1 || { echo "profile $wl failed"; tail -5 gpurun_out/$TAG/profile_$wl.log; exit 1; }
        echo "profile $wl done"
      done ;;
    bench)
      tools/bench_all.sh > gpurun_out/$TAG/bench_all.log 2>&1 || { echo "bench_all failed"; tail -5 gpurun_out/$TAG/bench_all.log; exit 1; }
      cat gpurun_out/$TAG/bench_all.log
      timeout -k 10 300 python3 bench.py --steps 200 --warmup 5 > gpurun_out/$TAG/bench200.json 2> gpurun_out/$TAG/bench200.err || exit 1
      timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench20.json 2> gpurun_out/$TAG/bench20.err || exit 1
      python3 tools/ggx_timing.py 2>/dev/null | tee gpurun_out/$TAG/ggx.log
      tools/ggx_trace.sh > gpurun_out/$TAG/ggx_trace.log 2>&1 || { echo "ggx trace failed"; tail -5 gpurun_out/$TAG/ggx_trace.log; exit 1; } ;;
    parts)
      GPU_MAX_HW_QUEUES=32 timeout -k 10 300 python3 tools/part_timing.py bunny 200 8 2>/dev/null | tee gpurun_out/$TAG/part_timing_8.log
      RT_SPAN=0 GPU_MAX_HW_QUEUES=32 timeout -k 10 300 python3 tools/part_timing.py bunny 200 8 2>/dev/null | tee gpurun_out/$TAG/part_timing_8_span.log
      GPU_MAX_HW_QUEUES=32 timeout -k 10 300 python3 tools/part_probe.py 300 8 2 4 8 2>/dev/null | tee gpurun_out/$TAG/part_probe_8.log ;;
    parity)
      timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q -s --timeout 300 --timeout-method thread -k "full_size" 2>&1 | grep -E "parity|passed|failed" | tee gpurun_out/$TAG/parity_full.log ;;
  esac
done
