#!/bin/bash
# Runs on the GPU box: per-kernel durations of the GGX test scene (1000x1000, 4 rays per pixel, depth 3) rendered by
# rt_render without statistics, launch by launch in the order of a frame.
set -o pipefail
OUT=gpurun_out/trace_ggx
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/ggx_timing.py trace > $OUT/run.txt 2>&1 || { tail -5 $OUT/run.txt; exit 1; }
python3 - $OUT <<'PY' | tee $OUT/launches.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-40s calls %5s  avg %9.1f us  total %9.1f us  %5s %%" % (r["Name"].replace("(anonymous namespace)::", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                      float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last frame: from the last camera kernel on
    last = max(i for i, r in enumerate(rows) if "k_camera_shade_packet" in r["Kernel_Name"])
    t0 = int(rows[last]["Start_Timestamp"])
    prev_end = t0
    for r in rows[last:]:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("   +%8.1f us  gap %6.1f  %-28s %8.1f us  grid %s" % ((a - t0) / 1e3, (a - prev_end) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:28],
                                                              (b - a) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
        prev_end = b
PY
find $OUT -name "*kernel_trace.csv" -delete
