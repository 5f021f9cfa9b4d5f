#!/bin/bash
set -e
O=gpurun_out/r02d
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined" > $O/pipe_test.log 2>&1 || { tail -40 $O/pipe_test.log; exit 1; }
tail -3 $O/pipe_test.log
for cfg in "p1 --pipeline 1" "p1_legacy --pipeline 1 --flags 8" "p2 --pipeline 2" "p2_legacy --pipeline 2 --flags 8" "p1_fused --pipeline 1 --flags 2" "p2_fused --pipeline 2 --flags 2" "p3_fused --pipeline 3 --flags 2"; do
  set -- $cfg; name=$1; shift
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json,sys; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
