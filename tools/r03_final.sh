#!/bin/bash
# Final checks of a binary, part B (part A = r03_final_tests.sh): the default bench line, the rocprofv3 kernel
# summaries (default two-slot schedule, single stream), the SQ counter pass behind roofline.achieved (profiles/pmc_valu.json) and the
# FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic (profiles/pmc_traffic.json).  Outputs under gpurun_out/r03final.
set -e
export TMPDIR=/tmp
O=gpurun_out/r03final
mkdir -p $O
python tools/gen_cache.py --pairs 64 > $O/gen.log 2>&1
B="python3 bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz"
K="k_linearize_lists<false"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o pmc -- $B --steps 2 --warmup 2 > $O/pmc_sq.log 2>&1
python tools/pmc_valu.py $O/pmc_sq "$K" 100000_1000000_64_GN $O/pmc_valu.json > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o pmc -- $B --steps 2 --warmup 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o pmc -- $B --steps 2 --warmup 2 > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 100000_1000000_64_GN $O/pmc_traffic.json "$K" > /dev/null
cp $O/pmc_valu.json $O/pmc_traffic.json profiles/    # so that the bench run below finds counters of THIS binary
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']; print('bench', round(d['value']), d['ms_per_step'], 'valu frac', r['frac'], 'alone', r.get('kernel_alone_frac'), 'hbm frac', r.get('hbm_frac_physical'), d['config']['parity_ok'], d['cpu_baseline']['value'], d['cpu_baseline']['one_thread']['value'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_default -o kt -- $B --steps 20 --warmup 3 > $O/kt_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_single -o kt -- $B --steps 20 --warmup 3 --pipeline 1 > $O/kt_single.log 2>&1
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
grep -o '"value": [0-9.]*' $O/kt_default.log $O/kt_single.log
