#!/bin/bash
set -e
O=gpurun_out/r02sched6
mkdir -p $O
python tools/gen_cache.py --pairs 128 --out /tmp/pcm_pairs128.npz > $O/gen.log 2>&1
run() { name=$1; shift
  python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs128.npz "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 0; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['roofline']['frac'],3), d['config']['mean_linearize_passes'])"; }
run p2_64 --pairs-per-gpu 64
run p2_96 --pairs-per-gpu 96
run p2_128 --pairs-per-gpu 128
run p2_64b --pairs-per-gpu 64
