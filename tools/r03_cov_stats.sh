#!/bin/bash
# diagnostic build (-DPCM_COV_STATS, tools/ab/): where the time of k_covariances goes -- passes, probe rounds, staged candidates, phase
# times per workgroup -- with the scan's index in sub-voxel order (default) and in input order
export TMPDIR=/tmp
O=gpurun_out/r03cov
mkdir -p $O
timeout -k 10 300 python tools/bench_gicp.py --cpu 0 --reps 1 --pairs 1 --models GICP --lib tools/ab/libpcm_amd_covstats.so > $O/fine.json 2> $O/fine.err; grep 'cov_' $O/fine.err
PCM_COV_SUBSORT=0 timeout -k 10 300 python tools/bench_gicp.py --cpu 0 --reps 1 --pairs 1 --models GICP --lib tools/ab/libpcm_amd_covstats.so > $O/coarse.json 2> $O/coarse.err; grep 'cov_' $O/coarse.err
