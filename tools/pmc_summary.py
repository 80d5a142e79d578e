"""Aggregate a rocprofv3 --pmc CSV (one row per dispatch and counter) into per-kernel averages."""
import csv
import glob
import sys
from collections import defaultdict


def main(paths, out):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for path in paths:
        for fn in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
            with open(fn) as f:
                rows = sorted(csv.DictReader(f), key=lambda r: int(r['Dispatch_Id']))
                seen = {}
                for row in rows:
                    k = row['Kernel_Name'].split('(')[0]
                    c = row['Counter_Name']
                    if k == 'lstm_layer_kernel':      # launches alternate layer 0, layer 1 (2-layer LSTM)
                        d = row['Dispatch_Id']
                        if d not in seen:
                            seen[d] = len(seen) % 2
                        k = 'lstm_layer_kernel[l%d]' % seen[d]
                    acc[k][c] += float(row['Counter_Value'])
                    cnt[k][c] += 1
    with open(out, 'w') as f:
        f.write('kernel,counter,dispatches,avg_per_dispatch,total\n')
        for k in sorted(acc):
            for c in sorted(acc[k]):
                f.write('%s,%s,%d,%.6g,%.6g\n' % (k, c, cnt[k][c], acc[k][c] / cnt[k][c], acc[k][c]))
    print(open(out).read())


if __name__ == '__main__':
    main(sys.argv[2:], sys.argv[1])
