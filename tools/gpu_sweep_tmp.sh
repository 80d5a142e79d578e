for s in 24576 28672 32768; do
timeout -k 10 300 python -u bench.py --cpu-seconds 0 --two-stage-ods 0 --steps 1 --slots $s > gpurun_out/r2n_slots$s.log 2>&1 || exit 1
done
