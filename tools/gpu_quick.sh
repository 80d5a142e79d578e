#!/bin/bash
# One GPU-box call while iterating: parity tests, then a short BJ-scale bench line (no CPU baseline, no two-stage).
# usage: tools/gpu_quick.sh TAG [bench args]
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${tag}_pytest.log 2>&1 || { tail -30 gpurun_out/${tag}_pytest.log; exit 1; }
tail -1 gpurun_out/${tag}_pytest.log
timeout -k 10 400 python -u bench.py --cpu-seconds 0 --two-stage-ods 0 --two-engines 0 --fp32-step 0 --steps 1 "$@" > gpurun_out/${tag}_bench.log 2>&1 || { tail -30 gpurun_out/${tag}_bench.log; exit 1; }
python - <<PY
import json
l=[x for x in open('gpurun_out/${tag}_bench.log') if x.startswith('{')][-1]
j=json.loads(l)
print('traj/s %.0f  ms/step %.0f  supersteps %d  page_waits %d  restarts %d' % (j['value'], j['ms_per_step'], j['supersteps'], j['page_waits'], j['restarts']))
for k,v in j['kernels'].items(): print('  %-24s %7.1f us  share %.3f  frac %.3f' % (k, v['avg_ms']*1e3, v['share'], v['frac']))
PY
python - <<PY
import json
l=[x for x in open('gpurun_out/${tag}_bench.log') if x.startswith('{')][-1]
g=json.loads(l).get('gat_encoder')
if g: print('  gat_encoder %.3f ms  frac %.3f' % (g['ms'], g['frac']))
PY
