#!/bin/bash
# kernel stats of the bench with the candidate-list kernel (default) and the tile kernel (flags 64)
export TMPDIR=/tmp
O=gpurun_out/r03nl
mkdir -p $O
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
for f in 0 64; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$f -o kt -- python3 bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --steps 10 --warmup 2 --flags $f > $O/kt$f.log 2>&1
echo "flags $f"; grep -o '"value": [0-9.]*' $O/kt$f.log | head -1; sort -t, -k3 -n -r $O/kt$f/*kernel_stats.csv | head -6 | cut -c1-200
done
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
