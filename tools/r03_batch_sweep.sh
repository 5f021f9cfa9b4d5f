#!/bin/bash
# bench.py over pairs per GPU and slots (candidate-list kernel): registrations/s and ms per step; 1 pair per step = the latency of one registration
export TMPDIR=/tmp
O=gpurun_out/r03sweep
mkdir -p $O
python tools/gen_cache.py --pairs 128 > $O/gen.log 2>&1
for cfg in "1 1" "8 1" "8 2" "32 1" "32 2" "64 1" "64 2" "128 1" "128 2"; do
  set -- $cfg; p=$1; s=$2
  timeout -k 10 400 python bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --pairs-per-gpu $p --pipeline $s --steps 40 --warmup 3 > $O/b_${p}_$s.json 2> $O/b_${p}_$s.err || tail -3 $O/b_${p}_$s.err
  python -c "import json; d=json.load(open('$O/b_${p}_$s.json')); print('pairs $p slots $s:', round(d['value']), 'reg/s', round(d['ms_per_step'],3), 'ms/step')"
done
