"""Sum the counters of every dispatch of the kernels whose name contains <substr> in a rocprofv3 --pmc output
directory: usage pmc_summary.py <dir> <substr> [out.json].  Prints per-launch averages."""
import csv, glob, json, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
sub = sys.argv[2]
acc = {}; disp = set()
for r in csv.DictReader(open(f)):
    if sub not in r["Kernel_Name"]:
        continue
    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp.add(r["Dispatch_Id"])
n = max(1, len(disp))
out = {"kernel_substr": sub, "launches": len(disp), "per_launch": {k: v / n for k, v in sorted(acc.items())}}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
