#!/bin/bash
set -e
O=gpurun_out/r02sched3
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
run() { name=$1; shift
  python bench.py --steps 24 --warmup 3 --cpu-seconds 0 "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), d['ms_per_step'])"; }
run split_p2 --schedule split --pipeline 2
run rot_p2 --pipeline 2
run rot_p3 --pipeline 3
run rot_p4 --pipeline 4
export PCM_BENCH_LAUNCH_EVENTS=0
run rot_p2_noev --pipeline 2
run rot_p3_noev --pipeline 3
run rot_p4_noev --pipeline 4
run rot_p3_w16_noev --pipeline 3 --window 16
run rot_p4_w12_noev --pipeline 4 --window 12
run rot_p6_noev --pipeline 6
