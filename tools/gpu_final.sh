#!/bin/bash
# Round-end evidence in one call: default bench line (with the CPU baseline), kernel stats, PMC passes.
set -o pipefail
tag=${1:-final}
timeout -k 10 600 python -u bench.py > gpurun_out/${tag}_bench_default.log 2>&1 || { tail -20 gpurun_out/${tag}_bench_default.log; exit 1; }
grep '^{' gpurun_out/${tag}_bench_default.log | tail -1 > gpurun_out/${tag}_bench_line.json
bash tools/gpu_profile.sh $tag
