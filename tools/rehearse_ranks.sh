set -o pipefail
mkdir -p gpurun_out/rehearsal
for n in 2 4 8; do
  RT_BENCH_BACKEND=gloo timeout -k 10 280 python3 bench.py --gpus $n --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/rehearsal/gloo_$n.json 2> gpurun_out/rehearsal/gloo_$n.err || { echo "gloo $n failed"; tail -5 gpurun_out/rehearsal/gloo_$n.err; exit 1; }
  python3 -c "import json; d=json.loads(open('gpurun_out/rehearsal/gloo_$n.json').read().strip().splitlines()[-1]); print($n, round(d['ms_per_step'],3), d['exchange'], d['config'].get('frame_check'), d['config']['parts_per_set_of_launches'], d.get('per_rank_ms',{}).get('render_ms'), round(d.get('value_bunny4k',0)/1e9,2))"
done
RT_BENCH_MULTI_PATH=1 timeout -k 10 280 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/rehearsal/rccl_1.json 2> gpurun_out/rehearsal/rccl_1.err || { echo "rccl 1 failed"; tail -5 gpurun_out/rehearsal/rccl_1.err; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/rehearsal/rccl_1.json').read().strip().splitlines()[-1]); print('rccl1', round(d['ms_per_step'],3), d['exchange'], d['rccl_version'], d['exchange_note'], d.get('per_rank_ms'))"
