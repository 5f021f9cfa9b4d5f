#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02l
mkdir -p $O
for p in 2 4; do
rocprofv3 --kernel-trace --output-format csv -d $O/kt$p -o kt -- python3 bench.py --steps 6 --warmup 3 --cpu-seconds 0 --pipeline $p --gen-workers 1 > $O/bench_kt$p.json 2> $O/bench_kt$p.err
python tools/trace_busy.py $O/kt$p 0.4 | sed "s/^/p$p: /"
done
find $O -name "*.csv" -size +20M -delete
