"""dev: do kernels of independent autoregressive chains on different streams overlap on the GPU?
Reads a rocprofv3 --kernel-trace CSV of `bench_aux.py --workload ar --batch B --streams G` and prints, for the steady part of the
run: kernels per queue, sum of kernel durations, length of the union of their intervals (time the GPU ran at least one kernel),
the time two or more kernels were resident at once, and the idle time between kernels.  usage: dev_ar_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']) for r in rows
      if 'ar_' in r['Kernel_Name']]
ks.sort()
if not ks:
    sys.exit('no ar_* kernels in the trace')
# steady part: drop the first and the last 10 % of the dispatches (warm-up call, graph capture, tails)
n = len(ks)
ks = ks[n // 10: n - n // 10]
t0, t1 = ks[0][0], max(e for _, e, _, _ in ks)
per_q = defaultdict(lambda: [0, 0])
for s, e, q, _ in ks:
    per_q[q][0] += 1
    per_q[q][1] += e - s
ev = []
for s, e, _, _ in ks:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy1 = busy2 = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1:
        busy1 += t - last
    if depth >= 2:
        busy2 += t - last
    depth += d
    last = t
span = t1 - t0
tot = sum(v[1] for v in per_q.values())
print('dispatches analysed: {} over {:.3f} ms on {} queue(s)'.format(len(ks), span / 1e6, len(per_q)))
for q, (c, d) in sorted(per_q.items()):
    print('  queue {}: {} kernels, {:.3f} ms of kernel time, average {:.2f} us'.format(q, c, d / 1e6, d / c / 1e3))
print('sum of kernel durations      {:.3f} ms = {:.2f} of the span'.format(tot / 1e6, tot / span))
print('>= 1 kernel resident (union) {:.3f} ms = {:.2f} of the span'.format(busy1 / 1e6, busy1 / span))
print('>= 2 kernels resident        {:.3f} ms = {:.2f} of the span'.format(busy2 / 1e6, busy2 / span))
print('no kernel resident (idle)    {:.3f} ms = {:.2f} of the span'.format((span - busy1) / 1e6, (span - busy1) / span))
by_name = defaultdict(lambda: [0, 0])
for s, e, _, nm in ks:
    k = nm.split('(')[0].split('::')[-1]
    by_name[k][0] += 1
    by_name[k][1] += e - s
for k, (c, d) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:6]:
    print('  {:32s} {:6d} x {:7.2f} us'.format(k[:32], c, d / c / 1e3))
