#!/bin/bash
set -e
O=gpurun_out/r02sched4
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
run() { name=$1; shift
  python bench.py --steps 30 --warmup 3 --cpu-seconds 0 "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['roofline']['frac'],3))"; }
run rot_p2 --pipeline 2
run rot_p2_st0 --pipeline 2 --stagger 0
run rot_p2_st1 --pipeline 2 --stagger 1.0
run rot_p2_prio --pipeline 2 --slot-priority 1
run rot_p2_64 --pipeline 2 --pairs-per-gpu 64
run rot_p2_64_w32 --pipeline 2 --pairs-per-gpu 64 --window 32
run rot_p3 --pipeline 3
PCM_BENCH_LAUNCH_EVENTS=0 run rot_p2_noev --pipeline 2
PCM_BENCH_LAUNCH_EVENTS=1 run rot_p2_allev --pipeline 2
