#!/bin/bash
# A/B of the in-tree build against tools/ab/libpcm_amd_prev.so with the default bench schedule (alternating, 3 runs each)
set -e
O=gpurun_out/r02abb
mkdir -p $O
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
for rep in 1 2 3; do
  for lib in prev new; do
    if [ $lib = prev ]; then export PCM_AMD_LIBRARY=$PWD/tools/ab/libpcm_amd_prev.so; else unset PCM_AMD_LIBRARY; fi
    python bench.py --steps 60 --warmup 3 --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz > $O/${lib}_$rep.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/${lib}_$rep.json')); print('$lib', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config'].get('parity_ok'))"
  done
done
