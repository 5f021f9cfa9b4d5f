#!/bin/bash
# Diagnostic build of the library (-DPCM_COV_STATS, built here into /tmp): where the time of k_covariances goes -- passes, probe
# rounds, staged candidates, phase times per workgroup -- for one scan, with the scan index in sub-voxel order (default) and in
# input order.  (The counters synchronise the stream: one object at a time only.)
export TMPDIR=/tmp
O=gpurun_out/r03cov
mkdir -p $O /tmp/pcm_covstats
S=pointcloud-slam_amd/csrc
for f in pcm_api kernels linearize_counted linearize_reforder ndt gicp pclndt preprocess gicp_bfgs voxel_hash; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPCM_COV_STATS -w -c $S/$f.hip -o /tmp/pcm_covstats/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Wl,--version-script=$S/exports.map -o /tmp/pcm_covstats/libpcm_amd_covstats.so /tmp/pcm_covstats/*.o || exit 1
timeout -k 10 300 python tools/bench_gicp.py --cpu 0 --reps 1 --pairs 1 --models GICP --lib /tmp/pcm_covstats/libpcm_amd_covstats.so > $O/subvoxel.json 2> $O/subvoxel.err; grep 'cov_' $O/subvoxel.err
PCM_COV_SUBSORT=0 timeout -k 10 300 python tools/bench_gicp.py --cpu 0 --reps 1 --pairs 1 --models GICP --lib /tmp/pcm_covstats/libpcm_amd_covstats.so > $O/input_order.json 2> $O/input_order.err; grep 'cov_' $O/input_order.err
