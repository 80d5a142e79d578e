#!/bin/bash
# Memory-side counters of the engine's kernels at full slot count: L1->L2 read latency, TLB misses, L2 hit rate.
set -o pipefail
tag=${1:-mem}
export TMPDIR=/tmp
mkdir -p gpurun_out
B="bench.py --ods 32768 --warmup 0 --cpu-seconds 0 --profile-every 0 --two-stage-ods 0"
rm -rf gpurun_out/${tag}_a gpurun_out/${tag}_b
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_a -- python3 $B > gpurun_out/${tag}_a.log 2>&1 || { tail -20 gpurun_out/${tag}_a.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_b -- python3 $B > gpurun_out/${tag}_b.log 2>&1 || { tail -20 gpurun_out/${tag}_b.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for d in ('gpurun_out/${tag}_a', 'gpurun_out/${tag}_b'):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r['Kernel_Name'].split('(')[0][:40], r['Counter_Name'])
            acc[k][0] += 1
            acc[k][1] += float(r['Counter_Value'])
    for (kn, cn), (n, v) in sorted(acc.items()):
        if any(s in kn for s in ('frontier', 'lstm_layer')):
            print('%-42s %-34s launches %6d  per launch %14.0f' % (kn, cn, n, v / n))
PY
rm -rf gpurun_out/${tag}_a gpurun_out/${tag}_b
