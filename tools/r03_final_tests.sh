#!/bin/bash
# Final check, part A: __graft_entry__.smoke and the whole GPU suite.  Outputs under gpurun_out/r03final.
set -e
export TMPDIR=/tmp
O=gpurun_out/r03final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
# the N>1 code path of bench.py (process group, per-pass all-gather on the collective thread) at world size 1: RCCL executes
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
timeout -k 10 300 python bench.py --collectives-at-one 1 --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --steps 30 --warmup 3 > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err || tail -5 $O/bench_rccl_world1.err
python -c "import json; d=json.loads(open('$O/bench_rccl_world1.json').read().strip().splitlines()[-1]); c=d['config']; print('rccl world 1:', round(d['value']), c['collectives_executed'], c['gathered_ok'], c['gather_ms_per_pass'], c['slot_wait_for_gather_ms_per_pass'])"
