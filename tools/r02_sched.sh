#!/bin/bash
# schedule sweep of bench.py on one box: slots x pairs per step x window (no cpu baseline leg)
set -e
O=gpurun_out/r02sched
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
run() { name=$1; shift
  python bench.py --steps 12 --warmup 3 --cpu-seconds 0 "$@" > $O/$name.json 2> $O/$name.err
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), d['ms_per_step'], d['config'].get('parity_ok'))"; }
run p2 --pipeline 2
run p4 --pipeline 4
run p8 --pipeline 8
run p8_64 --pipeline 8 --pairs-per-gpu 64
run p16_64 --pipeline 16 --pairs-per-gpu 64
run p4_64_w8 --pipeline 4 --pairs-per-gpu 64 --window 8
run p4_128_w12 --pipeline 4 --pairs-per-gpu 128 --window 12
run p2_64_w24 --pipeline 2 --pairs-per-gpu 64 --window 24
