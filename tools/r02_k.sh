#!/bin/bash
set -e
O=gpurun_out/r02k
mkdir -p $O
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
for m in 0 1 2 4 8 16 32; do
  PCM_FUSE_MAX_PAIRS=$m python tools/prof_single.py --pairs 32 --steps 10 --phases 0 --cache /tmp/pcm_pairs.npz 2>/dev/null | tail -1 | sed "s/^/fuse<=$m: /"
done
for m in 0 4 8; do
for p in 2 3; do
  PCM_FUSE_MAX_PAIRS=$m python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --pipeline $p > $O/bench_${m}_$p.json 2> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_${m}_$p.json')); print('bench fuse<=$m pipeline $p', round(d['value']), d['ms_per_step'])"
done; done
