#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02finab
mkdir -p $O
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
B="python3 bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --steps 20 --warmup 3"
for lib in cur alt; do
  [ $lib = alt ] && export PCM_AMD_LIBRARY=$PWD/tools/ab/libpcm_amd_alt.so
  for v in default single; do
    extra=""; [ $v = single ] && extra="--pipeline 1"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${lib}_$v -o kt -- $B $extra > $O/kt_${lib}_$v.log 2>&1
    python - <<PY
import csv
rows=list(csv.reader(open("$O/kt_${lib}_$v/kt_kernel_stats.csv")))
for r in rows[1:4]:
    if "finish" in r[0]: print("$lib $v", r[0][:25], r[1], r[3])
PY
    grep -o '"value": [0-9.]*' $O/kt_${lib}_$v.log
  done
done
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
