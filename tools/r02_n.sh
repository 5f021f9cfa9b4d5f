#!/bin/bash
set -e
O=gpurun_out/r02n
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "window" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
for cfg in "A --pairs-per-gpu 32 --pipeline 2" "B --pairs-per-gpu 64 --pipeline 1 --window 32" "C --pairs-per-gpu 64 --pipeline 2 --window 16" "D --pairs-per-gpu 64 --pipeline 2 --window 24" "E --pairs-per-gpu 96 --pipeline 1 --window 32" "F --pairs-per-gpu 128 --pipeline 2 --window 32" "G --pairs-per-gpu 128 --pipeline 2 --window 24 --slot-priority 1"; do
  set -- $cfg; name=$1; shift
  GPU_MAX_HW_QUEUES=8 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json; d=json.load(open('$O/bench_$name.json')); print('$name', '$*', round(d['value']), d['ms_per_step'], d['config']['mean_linearize_passes'], d['config']['converged'])"
done
