"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals, union busy time, pairwise overlap."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40],
                     r.get('Queue_Id', ''), r.get('Stream_Id', '')))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
tot = defaultdict(lambda: [0, 0])
for s, e, n, q, st in rows:
    tot[n][0] += e - s
    tot[n][1] += 1
busy, cur_s, cur_e = 0, None, None
for s, e, n, q, st in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('span ms', (t1 - t0) / 1e6, 'union busy ms', busy / 1e6, 'sum of kernels ms', sum(v[0] for v in tot.values()) / 1e6)
print('queues', sorted(set(r[3] for r in rows)), 'streams', sorted(set(r[4] for r in rows)))
for n, (d, c) in sorted(tot.items(), key=lambda x: -x[1][0])[:10]:
    print('%-42s n=%7d total %9.2f ms avg %8.2f us' % (n, c, d / 1e6, d / c / 1e3))
