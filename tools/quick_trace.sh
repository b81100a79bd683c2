#!/bin/bash
# Runs on the GPU box: per-kernel durations of one workload, every kernel alone on the GPU.
# Usage: tools/quick_trace.sh [workload]
set -o pipefail
WL=${1:-bunny}
OUT=gpurun_out/trace_$WL
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --workload $WL --flights 1 > $OUT/run.txt 2>&1 || { tail -5 $OUT/run.txt; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "at::native" in r["Name"]:
            continue
        print("%-60s calls %5s  avg %9.1f us  %5s %%" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
# kernels launched several times per frame (one per bounce): the mean duration of each launch position
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    by = {}
    for r in rows:
        by.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name, d in by.items():
        if "k_bounce" in name or "k_shade<false>" in name or "k_trace_fast" in name:
            per = 6
            tail = d[-per * 10:]
            print(name.replace("(anonymous namespace)::", "")[:40], "per position (us):", [round(sum(tail[k::per]) / len(tail[k::per]), 1) for k in range(per)])
PY
find $OUT -name "*kernel_trace.csv" -delete
