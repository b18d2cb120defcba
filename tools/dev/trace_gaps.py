"""Where does a block wait?  From a rocprofv3 --kernel-trace CSV of bench.py: kernels are grouped per hardware queue (one
block in flight = one HIP stream = one queue), ordered in time, and the gap between the end of one kernel and the start of the
next one on the same queue is summed per (previous kernel -> next kernel) pair over the steady state.
python tools/dev/trace_gaps.py <kernel_trace.csv>"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m = re.search(r'fx_\w+_kernel', n); return m.group(0) if m else n[:24]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id')) for r in rows)
t0 = ev[int(len(ev) * 0.55)][0]; t1 = ev[-40][0]
perq = collections.defaultdict(list)
for s, e, n, q in ev:
    if s >= t0 and e <= t1: perq[q].append((s, e, n))
gap = collections.defaultdict(float); cnt = collections.Counter(); dur = collections.defaultdict(float); dcnt = collections.Counter()
for q, lst in perq.items():
    lst.sort()
    for (s0, e0, n0), (s1, e1, n1) in zip(lst, lst[1:]):
        gap[(n0, n1)] += (s1 - e0) / 1e3; cnt[(n0, n1)] += 1
    for s, e, n in lst: dur[n] += (e - s) / 1e3; dcnt[n] += 1
nblk = dcnt.get('fx_walk_kernel', 1)
print("steady-state window %.1f ms, %d blocks (walker launches), %d queues" % ((t1 - t0) / 1e6, nblk, len(perq)))
print("per block: kernel time %.1f us, gaps %.1f us" % (sum(dur.values()) / nblk, sum(gap.values()) / nblk))
print("%-26s -> %-26s  mean gap us   per block us" % ("after", "before"))
for k, v in sorted(gap.items(), key=lambda kv: -kv[1])[:24]:
    print("%-26s -> %-26s  %10.1f   %10.1f" % (k[0], k[1], v / cnt[k], v / nblk))
print("kernel durations per block (us):", {n: round(v / nblk, 1) for n, v in sorted(dur.items(), key=lambda kv: -kv[1])})
