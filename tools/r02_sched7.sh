#!/bin/bash
set -e
O=gpurun_out/r02sched7
mkdir -p $O
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
run() { name=$1; shift
  python bench.py --steps 100 --warmup 5 --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), d['ms_per_step'], round(d['roofline']['frac'],3))"; }
run st050
run st025 --stagger 0.25
run st075 --stagger 0.75
run st035 --stagger 0.35
run st050b
