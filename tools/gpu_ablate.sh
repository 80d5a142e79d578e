#!/bin/bash
# Timing ablations in one GPU-box call: each line of the list is "TAG|LIB|bench args"; LIB = a library built by
# tools/build_variant.sh (ablate/libtsg_NAME.so) or "-" for the shipped one.  Prints one summary line per entry.
# usage: tools/gpu_ablate.sh 'tag1|-|--engines 2' 'tag2|ablate/libtsg_spw16.so|--engines 2' ...
set -o pipefail
mkdir -p gpurun_out
for spec in "$@"; do
  tag=${spec%%|*}; rest=${spec#*|}; lib=${rest%%|*}; args=${rest#*|}
  if [ "$lib" != "-" ]; then export TSG_LIB_OVERRIDE=$lib; else unset TSG_LIB_OVERRIDE; fi
  timeout -k 10 300 python -u bench.py --cpu-seconds 0 --two-stage-ods 0 --two-engines 0 --fp32-step 0 --steps 1 $args > gpurun_out/abl_${tag}.log 2>&1 || { echo "$tag FAILED"; tail -5 gpurun_out/abl_${tag}.log; exit 1; }
  python - <<PY
import json
l=[x for x in open('gpurun_out/abl_${tag}.log') if x.startswith('{')][-1]
j=json.loads(l)
ks=' '.join('%s=%.1f' % (k.replace('_kernel','').replace('lstm_layer','L'), v['avg_ms']*1e3) for k,v in j['kernels'].items())
print('${tag}: traj/s %.0f  supersteps %d  waits %d restarts %d  slots %d  | %s' % (j['value'], j['supersteps'], j['page_waits'], j['restarts'], j['config']['slots_in_flight'], ks))
PY
done
