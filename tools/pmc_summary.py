"""Aggregate rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel averages.

usage: pmc_summary.py OUT.csv OUT.json DIR [DIR ...] [--bench-log LOG_OF_THE_PMC_RUN]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  MI355X_MICROARCH.md (HBM section): on
gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) coalesced reads at 64 bytes, i.e.
reports half of their bytes; WRITE_SIZE is exact for 16-B-per-lane stores; other access widths are
uncalibrated.  The JSON therefore carries the raw figures and, for the kernels whose reads are 16 B per lane
(scoring, LSTM), `hbm_bytes_corrected` = 2 * fetch + write.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

WIDE_READERS = ('score_frontier_kernel', 'lstm_layer_split_kernel', 'lstm_layer0_split_cand_kernel', 'lstm_layer_kernel',
                'gat_aggregate_kernel')


def short(name):
    k = name.split('(')[0]
    if k.startswith('void '):
        k = k[5:]
    return k


def main(out_csv, out_json, paths, bench_line=None):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for path in paths:
        for fn in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
            with open(fn) as f:
                for row in csv.DictReader(f):
                    k = short(row['Kernel_Name'])
                    c = row['Counter_Name']
                    acc[k][c] += float(row['Counter_Value'])
                    cnt[k][c] += 1
    with open(out_csv, 'w') as f:
        f.write('kernel,counter,dispatches,avg_per_dispatch,total\n')
        for k in sorted(acc):
            for c in sorted(acc[k]):
                f.write('%s,%s,%d,%.6g,%.6g\n' % (k, c, cnt[k][c], acc[k][c] / cnt[k][c], acc[k][c]))
    js = {}
    for k in acc:
        if 'FETCH_SIZE' not in acc[k] or 'WRITE_SIZE' not in acc[k]:
            continue
        if k.startswith('at::') or k.startswith('__amd'):     # torch / runtime helpers: not kernels of the library
            continue
        fetch = acc[k]['FETCH_SIZE'] / cnt[k]['FETCH_SIZE'] * 1024.0
        write = acc[k]['WRITE_SIZE'] / cnt[k]['WRITE_SIZE'] * 1024.0
        e = {'dispatches': cnt[k]['FETCH_SIZE'], 'fetch_bytes_raw': fetch, 'write_bytes': write,
             'hbm_bytes_raw': fetch + write}
        if any(k.startswith(w) for w in WIDE_READERS):
            e['hbm_bytes_corrected'] = 2.0 * fetch + write
        js[k] = e
    if bench_line:      # work items per launch of the PMC pass (bench.py scales the traffic to its own launches)
        with open(bench_line) as f:
            lines = [x for x in f if x.startswith('{')]
        if lines:
            d = json.loads(lines[-1])
            if d.get('supersteps') and d.get('work_items'):
                js['_work_items_per_launch'] = d['work_items'] / d['supersteps']
    with open(out_json, 'w') as f:
        json.dump(js, f, indent=1, sort_keys=True)
    print(json.dumps(js, indent=1, sort_keys=True))


if __name__ == '__main__':
    args = sys.argv[3:]
    bl = None
    if '--bench-log' in args:
        i = args.index('--bench-log')
        bl = args[i + 1]
        args = args[:i] + args[i + 2:]
    main(sys.argv[1], sys.argv[2], args, bl)
