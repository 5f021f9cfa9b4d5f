#!/bin/bash
set -e
O=gpurun_out/r02b
mkdir -p $O
./tools/micro/valu_rate > $O/valu_rate.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 --pipeline 1 --cpu-seconds 0 > $O/bench_p1.json 2> $O/bench_p1.err
python bench.py --steps 20 --warmup 5 --pipeline 2 --cpu-seconds 0 > $O/bench_p2.json 2> $O/bench_p2.err
python bench.py --steps 20 --warmup 5 --pipeline 1 --flags 4 --cpu-seconds 0 > $O/bench_p1_sep.json 2> $O/bench_p1_sep.err
python bench.py --steps 20 --warmup 5 --pipeline 2 --flags 4 --cpu-seconds 0 > $O/bench_p2_sep.json 2> $O/bench_p2_sep.err
python bench.py --steps 20 --warmup 5 --pipeline 3 --cpu-seconds 0 > $O/bench_p3.json 2> $O/bench_p3.err
cat $O/valu_rate.txt
for f in p1 p2 p1_sep p2_sep p3; do python -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
