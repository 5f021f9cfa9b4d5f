#!/bin/bash
# the secondary configurations on the final binary: config 5 loop (two scan densities), GICP / VGICP, pclomp NDT (config 4)
export TMPDIR=/tmp
O=gpurun_out/r03sec
mkdir -p $O
timeout -k 10 300 python tools/bench_lio_loop.py --frames 40 > $O/lio_loop_raw_frames.json 2> $O/lio1.err || tail -3 $O/lio1.err
timeout -k 10 300 python tools/bench_lio_loop.py --frames 40 --leaf 0.5 > $O/lio_loop_leaf05.json 2> $O/lio2.err || tail -3 $O/lio2.err
timeout -k 10 300 python tools/bench_gicp.py --cpu 0 --reps 3 > $O/gicp.json 2> $O/gicp.err || tail -3 $O/gicp.err
timeout -k 10 400 python tools/bench_ndt.py --cpu 0 --reps 5 --scans 32 --models NDT_OMP,NDT_OMP_KDTREE > $O/ndt32.json 2> $O/ndt32.err || tail -3 $O/ndt32.err
timeout -k 10 400 python tools/bench_ndt.py --cpu 0 --reps 5 --scans 8 --models NDT_OMP,NDT_D2D,NDT_P2D > $O/ndt8.json 2> $O/ndt8.err || tail -3 $O/ndt8.err
python3 - <<'PY'
import json
O = "gpurun_out/r03sec/"
for f in ("lio_loop_raw_frames", "lio_loop_leaf05"):
    d = json.load(open(O + f + ".json")); print(f, round(d["hz_sustained"], 1), "Hz, frame ms", round(1e3 * d["frame"], 3), "match", round(1e3 * d["match"], 3))
for f in ("gicp", "ndt32", "ndt8"):
    d = json.load(open(O + f + ".json")); print(f, {k: (round(v["registrations_per_s"]), round(v["ms_per_batch"], 2)) for k, v in d.items()})
PY
