#!/bin/bash
# Runs on the GPU box: bench.py on every workload other than the default one (and the default one in
# reference-order mode), one JSON line each, collected into gpurun_out/other_workloads.json.
# Usage: tools/bench_all.sh
set -o pipefail
OUT=gpurun_out/other_workloads
mkdir -p $OUT
for spec in "spheres_1K fast" "cornell_room fast" "spheres_and_triangles fast" "simple_sphere_scene fast" "bunny4k fast" "bunny reference"; do
  set -- $spec
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload $1 --mode $2 2> $OUT/$1_$2.err | tail -n 1 > $OUT/$1_$2.json || { echo "$1 $2 failed"; tail -3 $OUT/$1_$2.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$OUT/$1_$2.json')); print('$1 $2', round(d['ms_per_step'],3), 'ms', round(d['value']/1e9,2), 'Grays/s')"
done
python3 - <<'PY'
import json, glob, os
out = {}
for f in sorted(glob.glob("gpurun_out/other_workloads/*.json")):
    out[os.path.basename(f)[:-5]] = json.load(open(f))
json.dump(out, open("gpurun_out/other_workloads.json", "w"), indent=1)
PY
