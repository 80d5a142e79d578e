#!/bin/bash
# One GPU-box call: parity tests, smoke, bench line, rocprofv3 kernel stats.  Each step under its own limit.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 200 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || { tail -30 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 400 python -u bench.py > gpurun_out/bench.log 2>&1 || { tail -30 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
rm -rf gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --warmup 0 --cpu-seconds 0 > gpurun_out/bench_prof.log 2>&1 || { tail -30 gpurun_out/bench_prof.log; exit 1; }
tail -1 gpurun_out/bench_prof.log
find gpurun_out/prof -name '*kernel_stats.csv' | head
# the full kernel trace is large: keep only the stats
find gpurun_out/prof -name '*kernel_trace.csv' -delete
