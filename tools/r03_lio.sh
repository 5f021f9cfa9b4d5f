#!/bin/bash
# One gpurun call: the sliding-map / LIO tests, the frame diagnostic (map update vs search) and the config-5 loop at two scan densities.
export TMPDIR=/tmp
O=gpurun_out/r03lio
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_lio.py tests/test_gpu_lio_frame.py tests/test_gpu_fullsize.py tests/test_gpu_edge.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 600 python tools/dbg_lio_slow.py 2>/dev/null | grep "^cap"
timeout -k 10 600 python tools/bench_lio_loop.py --frames 40 > $O/lio_loop_raw_frames.json 2> $O/lio_loop.err || tail -5 $O/lio_loop.err
timeout -k 10 600 python tools/bench_lio_loop.py --frames 40 --leaf 0.5 > $O/lio_loop_leaf05.json 2> $O/lio_loop2.err || tail -5 $O/lio_loop2.err
cat $O/lio_loop_raw_frames.json $O/lio_loop_leaf05.json
