#!/bin/bash
# GICP / VGICP after a change of the covariance pass: parity tests, then the bench without / with the sub-voxel order of the scan's
# kNN index, and with the optional fine index on top.
export TMPDIR=/tmp
O=gpurun_out/r03gicp
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_rbf.py tests/test_gicp_bfgs.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
PCM_COV_SUBSORT=0 timeout -k 10 300 python tools/bench_gicp.py --cpu 0 --reps 3 --models GICP,VGICP_direct1 > $O/gicp_input_order.json 2> $O/err1.log || tail -5 $O/err1.log
PCM_COV_FINE_INDEX=1 timeout -k 10 300 python tools/bench_gicp.py --cpu 0 --reps 3 --models GICP,VGICP_direct1 > $O/gicp_fine.json 2> $O/err3.log || tail -5 $O/err3.log
timeout -k 10 300 python tools/bench_gicp.py --cpu 0 --reps 3 > $O/gicp.json 2> $O/err2.log || tail -5 $O/err2.log
python3 - <<'PY'
import json
for f in ("gicp_input_order", "gicp_fine", "gicp"):
    d = json.load(open("gpurun_out/r03gicp/%s.json" % f))
    print(f, {k: (round(v["registrations_per_s"]), round(v["ms_per_batch"], 2), round(v["target_cov_build_s_incl_map"] * 1e3, 1), v["iterations"]) for k, v in d.items()})
PY
