#!/bin/bash
set -e
O=gpurun_out/r02o
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
python tools/prof_single.py --pairs 32 --steps 10 --phases 0 --cache /tmp/pcm_pairs.npz 2>/dev/null | tail -1 | sed "s/^/cache on : /"
python tools/prof_single.py --pairs 32 --steps 10 --phases 0 --flags 4 --cache /tmp/pcm_pairs.npz 2>/dev/null | tail -1 | sed "s/^/cache off: /"
for cfg in "A --pairs-per-gpu 32 --pipeline 2" "A_off --pairs-per-gpu 32 --pipeline 2 --flags 4" "D --pairs-per-gpu 64 --pipeline 2 --window 24" "D_off --pairs-per-gpu 64 --pipeline 2 --window 24 --flags 4"; do
  set -- $cfg; name=$1; shift
  GPU_MAX_HW_QUEUES=8 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/all.log 2>&1 || { tail -30 $O/all.log; exit 1; }
tail -2 $O/all.log
