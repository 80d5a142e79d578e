"""Aggregate rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel averages.

usage: pmc_summary.py OUT.csv OUT.json DIR [DIR ...]
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

WIDE_READERS = ('score_frontier_kernel', 'lstm_layer_split_kernel', 'lstm_layer_kernel', 'gat_aggregate_kernel')


def short(name):
    k = name.split('(')[0]
    if k.startswith('void '):
        k = k[5:]
    return k


def main(out_csv, out_json, paths):
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
        fetch = acc[k]['FETCH_SIZE'] / cnt[k]['FETCH_SIZE'] * 1024.0
        write = acc[k]['WRITE_SIZE'] / cnt[k]['WRITE_SIZE'] * 1024.0
        e = {'dispatches': cnt[k]['FETCH_SIZE'], 'fetch_bytes_raw': fetch, 'write_bytes': write,
             'hbm_bytes_raw': fetch + write}
        if any(k.startswith(w) for w in WIDE_READERS):
            e['hbm_bytes_corrected'] = 2.0 * fetch + write
        js[k] = e
    with open(out_json, 'w') as f:
        json.dump(js, f, indent=1, sort_keys=True)
    print(json.dumps(js, indent=1, sort_keys=True))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
