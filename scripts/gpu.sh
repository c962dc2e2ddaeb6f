#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (exit 3 = nothing charged).  usage: scripts/gpu.sh <timeout-s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
