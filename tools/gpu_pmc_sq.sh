#!/bin/bash
# One SQ counter pass over the engine's kernels (wave cycles, waiting, issuing): tools/gpu_pmc_sq.sh TAG -> gpurun_out/TAG_pmc_sq.csv
set -o pipefail
tag=${1:-sq}
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/${tag}_pmc_sq
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d gpurun_out/${tag}_pmc_sq -- python3 bench.py --ods 8192 --warmup 0 --cpu-seconds 0 --profile-every 0 --two-stage-ods 0 > gpurun_out/${tag}_pmc_sq.log 2>&1 || { tail -20 gpurun_out/${tag}_pmc_sq.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_sq.csv gpurun_out/${tag}_pmc_sq.json gpurun_out/${tag}_pmc_sq > /dev/null
rm -rf gpurun_out/${tag}_pmc_sq
grep -E "frontier_kernel|score_frontier|lstm_layer" gpurun_out/${tag}_pmc_sq.csv
