#!/bin/bash
set -e
O=gpurun_out/r02m
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -x -q > $O/t.log 2>&1 || { tail -20 $O/t.log; exit 1; }
tail -2 $O/t.log
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
python tools/prof_single.py --pairs 32 --steps 10 --phases 0 --cache /tmp/pcm_pairs.npz 2>/dev/null | tail -1
for cfg in "p2 --pipeline 2" "p2prio --pipeline 2 --slot-priority 1" "p3 --pipeline 3" "p3prio --pipeline 3 --slot-priority 1" "p4 --pipeline 4" "p4prio --pipeline 4 --slot-priority 1"; do
  set -- $cfg; name=$1; shift
  GPU_MAX_HW_QUEUES=8 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value']), d['ms_per_step'])"
done
