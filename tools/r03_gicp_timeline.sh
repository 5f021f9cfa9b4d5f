#!/bin/bash
# Timeline of one GICP batch (8 scans prepared by 8 host threads): which kernels overlap, where the device idles.
export TMPDIR=/tmp
O=gpurun_out/r03gtl
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 tools/bench_gicp.py --cpu 0 --reps 2 --models GICP > $O/gicp.json 2> $O/gicp.err || tail -5 $O/gicp.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03gtl/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cov = [r for r in rows if "k_covariances" in r["Kernel_Name"]]
# the last batch = the last 8 source-covariance launches and everything after the first of them
t0 = int(cov[-8]["Start_Timestamp"]) - 2_000_000
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
ev = []
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e))
    if e - s > 40_000:
        print("%8.3f ms  +%8.1f us  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]))
ev.sort()
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.3f ms, device busy (union of kernels) %.3f ms, sum of kernel times %.3f ms, launches %d" % ((ev[-1][1] - ev[0][0]) / 1e6, busy / 1e6, sum(e - s for s, e in ev) / 1e6, len(ev)))
PY
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
cat $O/gicp.json | python3 -c "import json,sys; d=json.load(sys.stdin); print({k:(round(v['registrations_per_s']), round(v['ms_per_batch'],2)) for k,v in d.items()})"
