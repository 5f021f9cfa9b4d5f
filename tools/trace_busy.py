"""GPU busy fraction from a rocprofv3 kernel trace: union of the kernel intervals over the span of the last `frac` of the trace,
per-queue kernel time, and the gaps between consecutive kernels of each queue (usage: trace_busy.py <dir> [frac])."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name'].split('(')[0][-40:]) for r in rows)
t0 = ev[0][0]; t1 = max(e[1] for e in ev)
lo = t1 - (t1 - t0) * frac
ev = [e for e in ev if e[0] >= lo]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, q, n in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e[1] for e in ev) - ev[0][0]
print("span %.2f ms  busy(union) %.2f ms = %.1f %%  sum of kernel time %.2f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, sum(e[1] - e[0] for e in ev) / 1e6))
perq = collections.defaultdict(list)
for e in ev: perq[e[2]].append(e)
for q, L in perq.items():
    gaps = [L[i + 1][0] - L[i][1] for i in range(len(L) - 1)]
    gaps = [g for g in gaps if g > 0]
    kt = sum(e[1] - e[0] for e in L)
    print("queue %s: %d kernels, kernel time %.2f ms, median gap %.1f us, mean gap %.1f us, gaps > 20 us: %d (%.2f ms)" % (
        q, len(L), kt / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, (sum(gaps) / len(gaps) / 1e3) if gaps else 0,
        sum(g > 20000 for g in gaps), sum(g for g in gaps if g > 20000) / 1e6))
byname = collections.Counter()
for s, e, q, n in ev: byname[n] += e - s
for n, t in byname.most_common(8): print("  %-42s %.2f ms" % (n, t / 1e6))
