#!/bin/bash
# pclomp NDT, KDTREE search on the neighbour-leaf lists: alone in a process and as the second model of a process (the streams of the
# two lock-step groups must not share a hardware queue)
export TMPDIR=/tmp
run() { timeout -k 10 300 python tools/bench_ndt.py --cpu 0 --reps 5 --scans 32 --models $1 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$2', {k:(round(v['registrations_per_s']), round(v['ms_per_batch'],2)) for k,v in d.items()})"; }
run NDT_OMP,NDT_OMP_KDTREE "both models in one process"
run NDT_OMP_KDTREE "kdtree alone"
