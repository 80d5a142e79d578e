timeout -k 10 900 python -u bench.py --cpu-seconds 0 --config stress --ods 32768 > gpurun_out/bench_stress.log 2>&1 || { tail -20 gpurun_out/bench_stress.log; exit 1; }
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/bench_stress.log") if x.startswith("{")][-1])
print("stress value", round(d["value"]), "exp/s", round(d["expansions_per_s"]), "slots", d["config"]["slots_in_flight"], "pool", d["config"]["state_pool_pages"], "supersteps", d["supersteps"], "rows", round(d["roofline"]["rows_per_launch"]), "waits", d["page_waits"], "restarts", d["restarts"], "exp/od", round(d["expansions_per_od"]), "reached", d["reached_frac"], d["gat_encoder"], {k:(round(v["avg_ms"]*1000,1)) for k,v in d["kernels"].items()})
PY
