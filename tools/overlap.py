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
if len(sys.argv) > 2:
    mid = len(rows) // 2
    base = rows[mid][0]
    for s, e, n, q, st in rows[mid:mid + int(sys.argv[2])]:
        print('q%s %-28s start %9.1f us  dur %7.1f us' % (q, n[:28], (s - base) / 1e3, (e - s) / 1e3))
runs, last, n = [], None, 0
for s, e, nme, q, st in rows:
    if q != last:
        if last is not None:
            runs.append(n)
        last, n = q, 0
    n += 1
runs.append(n)
import statistics
print('queue runs: count', len(runs), 'median length', statistics.median(runs), 'max', max(runs))
per_q = defaultdict(int)
for s, e, nme, q, st in rows:
    per_q[q] += e - s
print('busy per queue ms', {q: v / 1e6 for q, v in per_q.items()})
seen = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r['Kernel_Name'].split('(')[0][:40]
        if k not in seen:
            seen[k] = (r['LDS_Block_Size'], r['Scratch_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'],
                       r['Workgroup_Size_X'], r['Grid_Size_X'])
for k, v in seen.items():
    print('%-42s lds %s scratch %s vgpr %s agpr %s sgpr %s wg %s grid %s' % ((k,) + v))
