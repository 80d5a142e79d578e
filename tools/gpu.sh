#!/bin/bash
# gpurun wrapper for the build container: retries only while the pool reports "no box free" (exit 3).
# usage: tools/gpu.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
