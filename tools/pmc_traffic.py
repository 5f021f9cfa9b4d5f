"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/pmc_traffic.json:
HBM-side bytes per launch of the dominant kernel (k_linearize, timed variant).

FETCH_SIZE / WRITE_SIZE are in KiB of L2<->fabric traffic (Infinity-Cache hits
included).  gfx950 correction from the guide: FETCH_SIZE reads exactly half of a
wide coalesced 16-B/lane stream, so the read side is doubled; WRITE_SIZE is exact
for 16-B streaming stores.  This kernel's reads are mostly 16-B float4 gathers and
32-B slot reads, i.e. the calibrated shape, but it is a gather, so the corrected
figure is an upper estimate of the read side."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_key
KERNEL = sys.argv[5] if len(sys.argv) > 5 else 'k_linearize<false, false'

def load(d, counter):
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
    tot = 0.0; disp = set()
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"] or r["Counter_Name"] != counter:
            continue
        tot += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    return tot, len(disp)

fetch_kib, n1 = load(sys.argv[1], 'FETCH_SIZE')
write_kib, n2 = load(sys.argv[2], 'WRITE_SIZE')
key = sys.argv[3]
out = {"workload_key": key, "source_key": source_key(), "kernel": KERNEL, "launches": n1,
       "fetch_size_kib_per_launch_raw": fetch_kib / max(1, n1), "write_size_kib_per_launch": write_kib / max(1, n2),
       "hbm_bytes_per_launch": (2.0 * fetch_kib / max(1, n1) + write_kib / max(1, n2)) * 1024.0,
       "correction": "read side x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B), write side exact"}
json.dump(out, open(sys.argv[4], 'w'), indent=1)
print(out)
