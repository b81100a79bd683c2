#!/bin/bash
# Runs on the GPU box: the GPU test suite, then the default bench at 200 steps and, for every workload named
# on the command line, the bench of that workload.
# Usage: tools/check_gpu.sh [workload ...]
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1
rc=$?
tail -n 3 gpurun_out/t.log
[ $rc -eq 0 ] || exit $rc
for wl in bunny "$@"; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 5 --workload $wl 2> gpurun_out/b200_$wl.err | tail -n 1 > gpurun_out/b200_$wl.json || exit 1
  python3 - $wl <<'PY'
import json, sys
d = json.load(open("gpurun_out/b200_%s.json" % sys.argv[1]))
print(sys.argv[1], "ms/frame", round(d["ms_per_step"], 4), "Grays/s", round(d["value"] / 1e9, 2), "device-resident", round(d["ms_per_step_device_resident"], 4),
      "single", round(d["ms_per_step_single_frame"], 4), {k: round(v, 4) for k, v in d["kernel_ms_per_step"].items() if v},
      "solo dominant", round(d["roofline"]["avg_launch_ms"], 4) if d.get("roofline") else None)
PY
done
