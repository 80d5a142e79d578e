for p in 3 5 6 8; do
  TSG_MAX_POPS=$p timeout -k 10 300 python -u bench.py --cpu-seconds 0 --two-stage-ods 0 --two-engines 0 --fp32-step 0 --steps 1 > gpurun_out/abl_pops$p.log 2>&1 || { echo pops$p FAILED; tail -5 gpurun_out/abl_pops$p.log; exit 1; }
  python - <<PY
import json
l=[x for x in open('gpurun_out/abl_pops$p.log') if x.startswith('{')][-1]
j=json.loads(l)
ks=' '.join('%s=%.1f' % (k.replace('_kernel','').replace('lstm_layer','L'), v['avg_ms']*1e3) for k,v in j['kernels'].items())
print('pops$p: traj/s %.0f  supersteps %d  waits %d restarts %d | %s' % (j['value'], j['supersteps'], j['page_waits'], j['restarts'], ks))
PY
done
