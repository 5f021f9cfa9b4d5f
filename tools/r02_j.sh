#!/bin/bash
set -e
O=gpurun_out/r02j
mkdir -p $O
env | grep -i -E "rocp|preload" > $O/env.txt || true
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
cp pointcloud-slam_amd/libpcm_amd.so /tmp/new.so
for v in new prev new prev; do
  if [ $v = prev ]; then cp tools/ab/libpcm_amd_prev.so pointcloud-slam_amd/libpcm_amd.so; else cp /tmp/new.so pointcloud-slam_amd/libpcm_amd.so; fi
  python tools/prof_single.py --pairs 32 --steps 10 --phases 0 --cache /tmp/pcm_pairs.npz 2>/dev/null | tail -2 | sed "s/^/$v: /"
done
cp /tmp/new.so pointcloud-slam_amd/libpcm_amd.so
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']), d['ms_per_step'], d['config']['gen_s'])"
cat $O/env.txt
