#!/bin/bash
# GPU suite, default bench line, 2-rank gloo rehearsal of the N>1 control flow on one GPU
set -e
O=gpurun_out/r02full
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cat $O/bench_default.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo --pairs-per-gpu 16 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err || { tail -20 $O/bench_2rank_gloo.err; exit 1; }
cat $O/bench_2rank_gloo.json
