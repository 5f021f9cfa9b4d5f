#!/bin/bash
# One gpurun call: the default bench line (2 slots x 64 pairs) with the per-voxel candidate lists (flags 0, the default) and with
# the tile kernel (flags 64); optionally other flag sets as arguments.
export TMPDIR=/tmp
O=gpurun_out/r03bench
mkdir -p $O
FLAGS="${@:-0 64}"
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
for rep in 1 2; do
  for f in $FLAGS; do
    timeout -k 10 400 python bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --steps 50 --warmup 3 --flags $f > $O/bench_f${f}_$rep.json 2> $O/bench_f${f}_$rep.err || tail -5 $O/bench_f${f}_$rep.err
    python -c "import json; d=json.load(open('$O/bench_f${f}_$rep.json')); r=d['roofline']; print('flags=$f rep=$rep', round(d['value']), 'reg/s', round(d['ms_per_step'],3), 'ms/step; launch ms 2slots', round(r['avg_launch_ms_two_slots'],4), 'single', round(r['avg_launch_ms_single_stream'],4), 'sum/step', round(r['sum_launch_ms_over_step_ms'],2), 'parity', d['config'].get('parity_ok'), 'cold', round(d['config']['cold_registrations_per_s']))"
  done
done
