#!/bin/bash
# config-5 loop under rocprofv3: per-call durations of the frame path's kernels
export TMPDIR=/tmp
O=gpurun_out/r03lioprof
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 tools/bench_lio_loop.py --frames 6 > $O/run.json 2> $O/run.err || tail -5 $O/run.err
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03lioprof/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if "k_linearize" in r["Kernel_Name"] or "k_lio_finish" in r["Kernel_Name"]]
for r in sel[:40]:
    print("%-40s start %10.3f ms dur %9.1f us grid %s wg %s vgpr %s sgpr %s lds %s scratch %s" % (r["Kernel_Name"][:40], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
PY
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
