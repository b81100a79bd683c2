#!/bin/bash
# Runs on the GPU box: frames in flight with one shared copy of the scene (RT_BENCH_SHARE_SCENE=1, the default) against a
# copy per backend, and the number of frames in flight with the shared copy.
set -o pipefail
mkdir -p gpurun_out/share
run() {  # share flights workload
  RT_BENCH_SHARE_SCENE=$1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 5 --flights $2 --workload $3 2> gpurun_out/share/$1_$2_$3.err | tail -n 1 > gpurun_out/share/$1_$2_$3.json || { tail -3 gpurun_out/share/$1_$2_$3.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/share/$1_$2_$3.json')); print('share=$1 flights=$2 $3', round(d['ms_per_step'],4), round(d['ms_per_step_device_resident'],4))"
}
for round in 1 2; do
  for f in 3 4 5 6 8; do run 1 $f bunny || exit 1; done
  run 0 4 bunny || exit 1
  for f in 4 6; do run 1 $f cornell_room || exit 1; run 1 $f spheres_1K || exit 1; done
done
