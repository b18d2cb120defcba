"""Steady-state timeline summary of a rocprofv3 --kernel-trace CSV of bench.py: python tools/dev/trace_summary.py <kernel_trace.csv>"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m = re.search(r'fx_\w+_kernel', n); return m.group(0) if m else n[:30]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id')) for r in rows)
t0 = ev[int(len(ev) * 0.55)][0]; t1 = ev[-20][0]
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
span = (t1 - t0) / 1e6
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
by = collections.defaultdict(list)
for s, e, n, q in sel: by[n].append((s, e))
nblk = len(by['fx_walk_kernel'])
print("window %.1f ms, %d blocks, %.3f ms per block" % (span, nblk, span / nblk))
for n, iv in by.items():
    print("%-24s n=%3d avg %.3f ms  busy (union) %.0f%% of window, mean concurrency %.2f" % (n, len(iv), sum(e - s for s, e in iv) / 1e6 / len(iv), 100 * union(iv) / 1e6 / span, sum(e - s for s, e in iv) / 1e6 / span))
perq = collections.defaultdict(list)
for s, e, n, q in sel: perq[q].append((s, e))
print("per hardware queue busy:", {q: "%.0f%%" % (100 * union(iv) / 1e6 / span) for q, iv in sorted(perq.items())})
