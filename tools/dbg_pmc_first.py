"""Diagnostic: counters of the first k_linearize dispatch in a rocprofv3 --pmc output directory."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = {}
for r in csv.DictReader(open(f)):
    if "k_linearize" not in r["Kernel_Name"]:
        continue
    k = (int(r["Dispatch_Id"]), r["Counter_Name"])
    acc[k] = acc.get(k, 0) + float(r["Counter_Value"])
first = min(d for d, _ in acc)
print(sys.argv[2] if len(sys.argv) > 2 else "", {c: v for (d, c), v in sorted(acc.items()) if d == first})
