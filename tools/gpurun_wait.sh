#!/bin/bash
# gpurun_wait.sh TIMEOUT 'command': gpurun, retried while the pool has no free slot (exit 3: nothing ran, nothing was charged)
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
