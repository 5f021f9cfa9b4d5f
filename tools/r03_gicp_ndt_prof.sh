#!/bin/bash
# GICP / VGICP and pclomp-NDT batches under rocprofv3: where the time of a batch goes (kernel sums vs wall).
export TMPDIR=/tmp
O=gpurun_out/r03gn
mkdir -p $O
summ() {   # $1 = trace dir, $2 = label
python3 - "$1" "$2" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0.0, []])
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[k][0] += 1; agg[k][1] += d; agg[k][2].append(d)
tot = sum(v[1] for v in agg.values())
print("== %s: %d launches, %.1f ms of kernel time" % (sys.argv[2], len(rows), tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %-70s n %6d  sum %9.1f us  avg %8.1f us  p50 %8.1f  %5.1f %%" % (k, v[0], v[1], v[1] / v[0], sorted(v[2])[len(v[2]) // 2], 100 * v[1] / tot))
if sys.argv[2] == "ndt":   # the rounds of the last batch: pass / step durations in launch order
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "k_pclndt_batch" in r["Kernel_Name"]]
    seq = seq[-(len(seq) // 4):]   # cold batch + 3 timed ones
    print("  last batch, us per round (pass/step):", " ".join("%.0f/%.0f" % (seq[i][1], seq[i + 1][1]) for i in range(0, len(seq) - 1, 2)))
PY
}
WHAT=${1:-both}
if [ $WHAT != ndt ]; then
rocprofv3 --kernel-trace --output-format csv -d $O/gicp -o kt -- python3 tools/bench_gicp.py --cpu 0 --reps 3 --models GICP > $O/gicp.json 2> $O/gicp.err || tail -5 $O/gicp.err
summ $O/gicp gicp > $O/gicp_kernels.txt; cat $O/gicp_kernels.txt
fi
if [ $WHAT != gicp ]; then
rocprofv3 --kernel-trace --output-format csv -d $O/ndt -o kt -- python3 tools/bench_ndt.py --cpu 0 --reps 3 --scans 32 --models NDT_OMP > $O/ndt.json 2> $O/ndt.err || tail -5 $O/ndt.err
summ $O/ndt ndt > $O/ndt_kernels.txt; cat $O/ndt_kernels.txt
fi
python3 -c "
import json
import os
for f in ('gicp','ndt'):
    if not os.path.exists('$O/%s.json' % f): continue
    d = json.load(open('$O/%s.json' % f))
    print({k: (round(v['registrations_per_s']), round(v['ms_per_batch'], 2)) for k, v in d.items()})
"
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
