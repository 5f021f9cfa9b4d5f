#!/bin/bash
# pclomp NDT after a change of the passes: parity tests, then config 4 at 32 and 8 scans.
export TMPDIR=/tmp
O=gpurun_out/r03ndt
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pclndt.py tests/test_gpu_ndt.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "ndt tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/bench_ndt.py --cpu 0 --reps 5 --scans 32 --models NDT_OMP,NDT_OMP_KDTREE > $O/ndt32.json 2> $O/err1.log || tail -5 $O/err1.log
timeout -k 10 400 python tools/bench_ndt.py --cpu 0 --reps 5 --scans 8 --models NDT_OMP > $O/ndt8.json 2> $O/err2.log || tail -5 $O/err2.log
python3 - <<'PY'
import json
for f in ("ndt32", "ndt8"):
    d = json.load(open("gpurun_out/r03ndt/%s.json" % f))
    print(f, {k: (round(v["registrations_per_s"]), round(v["ms_per_batch"], 2), v["iterations"][:8], v["err_vs_gt_m"][:4]) for k, v in d.items()})
PY
