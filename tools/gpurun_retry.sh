#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'   -- gpurun, retried every 90 s while no GPU slot / box is free (exit code 3)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
