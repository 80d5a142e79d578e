#!/bin/bash
# timing-only ablation builds of the scoring kernel (outputs are wrong on purpose)
for a in ${ABL:-1 2 3}; do
  TSG_LIB_OVERRIDE=$PWD/ablate/libtsg_ab$a.so timeout -k 10 200 python -u bench.py --ods 32768 --warmup 1 --cpu-seconds 0 --streams 1 > gpurun_out/ab$a.log 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open('gpurun_out/ab$a.log') if x.startswith('{')][-1]
d=json.loads(l)
print('ablate $a', 'value', round(d['value']), {k: round(v['avg_ms']*1000,1) for k,v in d['kernels'].items()})
PY
done
