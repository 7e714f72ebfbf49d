#!/bin/bash
# scripts/gpu.sh TIMEOUT 'command' - one gpurun call, retried while the pod's GPU slots are busy
# (exit code 3: nothing charged); creates gpurun_out/r04 on the box first (it is scratch there)
t=$1; shift
for attempt in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "mkdir -p gpurun_out/r04 && $*"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
