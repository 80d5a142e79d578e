#!/bin/bash
# Profiles of one round: kernel stats + the two PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/gpu_profile.sh TAG      -> gpurun_out/TAG_*
set -o pipefail
tag=${1:-prof}
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH="bench.py --warmup 0 --cpu-seconds 0 --two-stage-ods 0 --two-engines 0 --fp32-step 0"   # the default workload and timed step of the bench line, without the warm-up step and the two-stage sub-record (their small launches of the same kernels would enter the per-kernel averages) and without the CPU baseline
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 $BENCH > gpurun_out/${tag}_bench_stats.log 2>&1 || { tail -20 gpurun_out/${tag}_bench_stats.log; exit 1; }
find gpurun_out/${tag}_stats -name '*kernel_trace.csv' -delete
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py --ods 8192 --warmup 0 --cpu-seconds 0 --two-stage-ods 0 --two-engines 0 --fp32-step 0 --profile-every 0 > gpurun_out/${tag}_pmc_fetch.log 2>&1 || { tail -20 gpurun_out/${tag}_pmc_fetch.log; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py --ods 8192 --warmup 0 --cpu-seconds 0 --two-stage-ods 0 --two-engines 0 --fp32-step 0 --profile-every 0 > gpurun_out/${tag}_pmc_write.log 2>&1 || { tail -20 gpurun_out/${tag}_pmc_write.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_traffic.csv gpurun_out/${tag}_pmc_traffic.json gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write --bench-log gpurun_out/${tag}_pmc_fetch.log > /dev/null
rm -rf gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write
grep "^{" gpurun_out/${tag}_bench_stats.log | tail -1 > gpurun_out/${tag}_bench_line_profiled_run.json
cat gpurun_out/${tag}_pmc_traffic.json
