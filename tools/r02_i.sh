#!/bin/bash
set -e
O=gpurun_out/r02i
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']), d['ms_per_step'], {k:v for k,v in d['config'].items() if k.startswith('parity')}, d['cpu_baseline'])"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --pairs-per-gpu 8 --cpu-seconds 0 > $O/bench_gloo2.json 2> $O/bench_gloo2.err
python -c "import json; d=json.load(open('$O/bench_gloo2.json')); print('gloo2', round(d['value']), d['config']['gathered_ok'])"
