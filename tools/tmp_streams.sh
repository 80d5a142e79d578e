timeout -k 5 60 ./ablate/overlap_test && for st in 1 2; do timeout -k 10 300 python -u bench.py --cpu-seconds 0 --streams $st --ods 32768 > gpurun_out/st_$st.log 2>&1 && python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/st_$st.log") if x.startswith("{")][-1])
print("streams $st", round(d["value"]), d["supersteps"], {k:(round(v["avg_ms"]*1000,1), round(v["frac"],3)) for k,v in d["kernels"].items()})
PY
done
