#!/bin/bash
# Secondary configurations and the SQ counter passes on the round's final binary.
set -e
export TMPDIR=/tmp
O=gpurun_out/r02sec
mkdir -p $O
python tools/bench_lio_loop.py > $O/lio_loop_config5.json 2> $O/lio.err || tail -5 $O/lio.err
echo "lio done"
python tools/bench_gicp.py --cpu 0 > $O/gicp_vgicp.json 2> $O/gicp.err || tail -5 $O/gicp.err
echo "gicp done"
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
pass() { name=$1; shift; L="$@";
  rocprofv3 --pmc $L --kernel-trace --output-format csv -d $O/$name -o $name -- python3 tools/prof_single.py --pairs 32 --steps 1 --phases 0 --cache /tmp/pcm_pairs.npz > $O/$name.log 2>&1
  python tools/pmc_summary.py $O/$name k_linearize $O/$name.json > /dev/null; }
pass pmcA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass pmcB SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD
pass pmcD SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
python tools/prof_single.py --pairs 32 --steps 10 --phases 1 --cache /tmp/pcm_pairs.npz > $O/phases.log 2>&1; tail -3 $O/phases.log
