#!/bin/bash
# Queue helper: run one gpurun call; when no GPU slot is free (exit 3: nothing ran, nothing charged) wait and ask again.
# A call that RAN (any other exit code) is never repeated.   usage: tools/gpu.sh <timeout-seconds> '<command>'
set -o pipefail
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
