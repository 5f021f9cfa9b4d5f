#!/bin/bash
set -e
O=gpurun_out/r02f
mkdir -p $O
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --pipeline 1 --flags 8 > $O/legacy.json 2> $O/legacy.err
python -c "import json,sys; d=json.load(open('$O/legacy.json')); print('legacy', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"
for L in 1 2 3 4 6 8 13; do
  PCM_PIPE_TILES=$L python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --pipeline 1 > $O/L$L.json 2> $O/L$L.err
  python -c "import json,sys; d=json.load(open('$O/L$L.json')); print('L=$L', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
