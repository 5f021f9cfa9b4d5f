#!/bin/bash
set -e
O=gpurun_out/r02ab2
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
export GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
python bench.py --steps 40 --warmup 3 --cpu-seconds 0 > $O/bench_$rep.json 2> $O/bench_$rep.err
python -c "import json; d=json.load(open('$O/bench_$rep.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['roofline']['frac'],3))"
done
