#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02fin
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
B="python3 bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --steps 20 --warmup 3"
for v in default single; do
  extra=""; [ $v = single ] && extra="--pipeline 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- $B $extra > $O/kt_$v.log 2>&1
  python - <<PY
import csv
rows=list(csv.reader(open("$O/kt_$v/kt_kernel_stats.csv")))
for r in rows[1:4]: print("$v", r[0][:45], r[1], r[3])
PY
  grep -o '"value": [0-9.]*' $O/kt_$v.log
done
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
