#!/bin/bash
# Profiles of the final binary: rocprofv3 kernel summaries (default two-slot schedule and single stream) and the
# FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic (separate passes, program directly after `--`).
set -e
export TMPDIR=/tmp
O=gpurun_out/r02prof
mkdir -p $O
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
B="python3 bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_default -o kt -- $B --steps 10 --warmup 3 > $O/kt_default.log 2>&1
echo "kt_default done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_single -o kt -- $B --steps 10 --warmup 3 --pipeline 1 > $O/kt_single.log 2>&1
echo "kt_single done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pmc -- $B --steps 2 --warmup 2 > $O/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pmc -- $B --steps 2 --warmup 2 > $O/pmc_write.log 2>&1
echo "write done"
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 100000_1000000_64_GN $O/pmc_traffic.json
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
tail -2 $O/kt_default.log; tail -2 $O/kt_single.log
ls -la $O $O/kt_default/* | head -30
