#!/bin/bash
# One gpurun call: parity tests of the search kernels, an A/B on one stream (32 cached pairs) with phase stamps, and the default
# bench line (2 slots x 64 pairs) for both kernels.
export TMPDIR=/tmp
O=gpurun_out/r03ab
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_counted_search.py tests/test_gpu_reforder.py tests/test_gpu_parity.py tests/test_gpu_adapter_runs.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
for rep in 1 2; do
  for f in 8 0; do
    timeout -k 10 300 python tools/prof_single.py --pairs 32 --steps 10 --phases 1 --cache /tmp/pcm_pairs.npz --flags $f > $O/f${f}_$rep.log 2>&1
    echo "flags=$f rep=$rep: $(grep 'ms per' $O/f${f}_$rep.log)"; grep "ticks" $O/f${f}_$rep.log
  done
done
for rep in 1 2; do
  for f in 8 0; do
    timeout -k 10 400 python bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --steps 50 --warmup 3 --flags $f > $O/bench_f${f}_$rep.json 2> $O/bench_f${f}_$rep.err || tail -5 $O/bench_f${f}_$rep.err
    python -c "import json; d=json.load(open('$O/bench_f${f}_$rep.json')); r=d['roofline']; print('bench flags=$f rep=$rep', round(d['value']), 'reg/s', round(d['ms_per_step'],3), 'ms/step; launch ms 2slots', round(r['avg_launch_ms_two_slots'],4), 'single', round(r['avg_launch_ms_single_stream'],4))"
  done
done
