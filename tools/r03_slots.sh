export TMPDIR=/tmp
O=gpurun_out/r03slots
mkdir -p $O
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
for s in 2 3 4; do
  timeout -k 10 400 python bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --steps 50 --warmup 3 --pipeline $s > $O/bench_s$s.json 2> $O/bench_s$s.err || tail -5 $O/bench_s$s.err
  python -c "import json; d=json.load(open('$O/bench_s$s.json')); r=d['roofline']; print('slots=$s', round(d['value']), 'reg/s', round(d['ms_per_step'],3), 'ms/step; launch ms', round(r['avg_launch_ms_two_slots'],4), 'single', round(r['avg_launch_ms_single_stream'],4), 'sum/step', round(r['sum_launch_ms_over_step_ms'],2))"
done
