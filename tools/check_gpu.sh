#!/bin/bash
# Runs on the GPU box: the GPU test suite, then the default bench at 200 steps and the per-kernel times.
# Usage: tools/check_gpu.sh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1
rc=$?
tail -n 3 gpurun_out/t.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 200 --warmup 5 2> gpurun_out/b200.err | tail -n 1 > gpurun_out/b200.json || exit 1
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/b200.json"))
print("ms/frame", round(d["ms_per_step"], 4), "Grays/s", round(d["value"] / 1e9, 2))
for k in ("kernel_ms", "kernels", "roofline_issue"):
    if k in d:
        print(k, d[k])
PY
