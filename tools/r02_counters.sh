#!/bin/bash
# One gpurun call: single-stream kernel summary + SQ counter passes of k_linearize + phase stamps.
set -e
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 tools/prof_single.py --pairs 32 --steps 5 --phases 0 --cache /tmp/pcm_pairs.npz > $O/kt.log 2>&1
have() { grep -qw "$1" $O/counters_list.txt; }
pass() { name=$1; shift; L="$@";
  rocprofv3 --pmc $L --kernel-trace --output-format csv -d $O/$name -o $name -- python3 tools/prof_single.py --pairs 32 --steps 1 --phases 0 --cache /tmp/pcm_pairs.npz > $O/$name.log 2>&1
  python tools/pmc_summary.py $O/$name k_linearize $O/$name.json > /dev/null; }
pass pmcA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass pmcB SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD
pass pmcC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM
pass pmcD SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
ls -la $O
