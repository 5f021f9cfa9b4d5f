#!/bin/bash
# A/B of two builds of the library on the same box: one-stream batched aligns of 32 cached pairs
set -e
O=gpurun_out/r02ab
mkdir -p $O
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
for rep in 1 2; do
  python tools/prof_single.py --pairs 32 --steps 10 --phases 1 --cache /tmp/pcm_pairs.npz --lib tools/ab/libpcm_amd_prev.so > $O/prev_$rep.log 2>&1
  python tools/prof_single.py --pairs 32 --steps 10 --phases 1 --cache /tmp/pcm_pairs.npz > $O/new_$rep.log 2>&1
done
grep -H "ms per\|ticks" $O/prev_*.log $O/new_*.log
