#!/bin/bash
# Runs on the GPU box: two builds of the library against each other -- first that they render the same bits
# (tools/ab_bits.py), then the 200-frame bench of three workloads, A B A B.
# Usage: tools/ab_run.sh <a.so> [<b.so>]   (b defaults to the in-tree build; keep variants under variants/)
set -o pipefail
mkdir -p gpurun_out/ab
A=$(realpath $1)
B=$(realpath ${2:-rust-raytracer_amd/csrc/librt_hip.so})
RT_HIP_LIB=$A timeout -k 10 200 python3 tools/ab_bits.py gpurun_out/ab/a.npz > gpurun_out/ab/bits_a.log 2>&1 || { tail -5 gpurun_out/ab/bits_a.log; exit 1; }
RT_HIP_LIB=$B timeout -k 10 200 python3 tools/ab_bits.py gpurun_out/ab/b.npz > gpurun_out/ab/bits_b.log 2>&1 || { tail -5 gpurun_out/ab/bits_b.log; exit 1; }
python3 tools/ab_bits.py --compare gpurun_out/ab/a.npz gpurun_out/ab/b.npz > gpurun_out/ab/compare.log 2>&1; echo "compare rc=$?"; grep -c identical gpurun_out/ab/compare.log; grep DIFFERENT gpurun_out/ab/compare.log
rm -f gpurun_out/ab/a.npz gpurun_out/ab/b.npz
for round in 1 2; do for v in A B; do
  lib=$A; [ $v = B ] && lib=$B
  for wl in cornell_room bunny spheres_1K; do
    RT_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 5 --workload $wl 2> gpurun_out/ab/${v}_$wl.err | tail -n 1 > gpurun_out/ab/${v}_$wl.json || { tail -3 gpurun_out/ab/${v}_$wl.err; exit 1; }
    python3 -c "import json; d=json.load(open('gpurun_out/ab/${v}_$wl.json')); print('$v $wl', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['kernel_ms_per_step'].items() if x})"
  done
done; done
