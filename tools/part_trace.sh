#!/bin/bash
# Runs on the GPU box: kernel trace of one part of the bunny frame rendered repeatedly.
# Usage: tools/part_trace.sh <parts> <part> <frames> <flights> [hw queues]
set -o pipefail
OUT=gpurun_out/part_trace_$1_$4
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$5" ] && export GPU_MAX_HW_QUEUES=$5
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/part_trace.py run $1 $2 $3 $4 > $OUT/run.txt 2>&1 || { tail -5 $OUT/run.txt; exit 1; }
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 tools/part_trace.py read $f | tee $OUT/summary.txt
rm -f $f
