#!/bin/bash
# gpurun with retries while the pool reports "no box / slot free" (exit code 3: nothing charged). Usage: gpurun_retry.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
