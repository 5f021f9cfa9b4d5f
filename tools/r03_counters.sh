#!/bin/bash
# One gpurun call: SQ counter passes of the search kernels (k_linearize_lists via PCM_FLAG_NEIGHBOUR_LISTS = 16, the tile kernel
# k_linearize via PCM_FLAG_NO_NEIGHBOUR_LISTS = 64) on one stream, 32 cached bench pairs; per-launch averages ->
# gpurun_out/r03pmc/*.json (copy what is quoted into profiles/).
export TMPDIR=/tmp
O=gpurun_out/r03pmc
mkdir -p $O
python tools/gen_cache.py --pairs 32 > $O/gen.log 2>&1
pass() { name=$1; flags=$2; sub=$3; shift 3; L="$@";
  rocprofv3 --pmc $L --kernel-trace --output-format csv -d $O/$name -o $name -- python3 tools/prof_single.py --pairs 32 --steps 1 --phases 0 --cache /tmp/pcm_pairs.npz --flags $flags > $O/$name.log 2>&1
  python tools/pmc_summary.py $O/$name "$sub" $O/$name.json | tr -d '\n' | cut -c1-900; echo; }
for v in "16 k_linearize_lists<" "64 k_linearize<"; do
  set -- $v; f=$1; sub=$2
  pass A_f$f $f "$sub" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  pass B_f$f $f "$sub" SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
  pass C_f$f $f "$sub" FETCH_SIZE
  pass D_f$f $f "$sub" WRITE_SIZE
done
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
