#!/bin/bash
set -e
O=gpurun_out/r02sched2
mkdir -p $O
export GPU_MAX_HW_QUEUES=8
run() { name=$1; shift
  python bench.py --steps 12 --warmup 3 --cpu-seconds 0 "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), d['ms_per_step'])"; }
run p2_ev --pipeline 2
run p4_ev --pipeline 4
export PCM_BENCH_LAUNCH_EVENTS=0
run p2_noev --pipeline 2
run p4_noev --pipeline 4
run p4_noev_nostag --pipeline 4 --stagger 0
run p8_noev_nostag --pipeline 8 --stagger 0
