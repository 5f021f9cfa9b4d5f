#!/bin/bash
set -e
O=gpurun_out/r02sched5
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
run() { name=$1; shift
  python bench.py --steps 20 --warmup 3 --cpu-seconds 0 "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['roofline']['frac'],3), d['config']['mean_linearize_passes'])"; }
run rot_p2_64 --pipeline 2 --pairs-per-gpu 64
run rot_p2_64_st0 --pipeline 2 --pairs-per-gpu 64 --stagger 0
run rot_p3_64 --pipeline 3 --pairs-per-gpu 64
run rot_p2_48 --pipeline 2 --pairs-per-gpu 48
run rot_p2_128_w64 --pipeline 2 --pairs-per-gpu 128 --window 64
run rot_p2_96_w64 --pipeline 2 --pairs-per-gpu 96 --window 64
