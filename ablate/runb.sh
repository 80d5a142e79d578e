for st in 1 2 4; do timeout -k 10 300 python -u bench.py --cpu-seconds 0 --streams $st > gpurun_out/b6_$st.log 2>&1; python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/b6_$st.log") if x.startswith("{")][-1])
print("streams $st", round(d["value"]), {k:(round(v["avg_ms"]*1000,1), round(v["frac"],3)) for k,v in d["kernels"].items()})
PY
done
