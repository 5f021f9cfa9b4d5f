"""Print the per-round kernel timeline of one batched align from a rocprofv3 kernel trace CSV
(usage: trace_rounds.py <dir> [index of the k_init_states launch, default -2])."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_init_states' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
start = idx[which]; end = idx[which + 1] if which + 1 < 0 else len(rows)
sel = rows[start:end]
t0 = int(sel[0]['Start_Timestamp']); prev = None; tot = {}
for r in sel:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    full = r['Kernel_Name']
    n = 'k_linearize' if 'k_linearize' in full else full.split('(')[0][-26:]
    tot[n] = tot.get(n, 0) + (e - s) / 1e3
    if any(k in n for k in ('k_linearize', 'k_finish', 'k_trial')):
        print("%8.1f dur %6.1f gap %5.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0, n))
    prev = e
print({k: round(v, 1) for k, v in tot.items()})
print("span us", (int(sel[-1]['End_Timestamp']) - t0) / 1e3)
