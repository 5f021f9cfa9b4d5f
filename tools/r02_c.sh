#!/bin/bash
set -e
O=gpurun_out/r02c
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
for cfg in "p1 --pipeline 1" "p2 --pipeline 2" "p2_sep --pipeline 2 --flags 4" "p1_sep --pipeline 1 --flags 4" "p4 --pipeline 4"; do
  set -- $cfg; name=$1; shift
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
