#!/bin/bash
# gpu_step.sh SECONDS cmd...: one GPU step under its own timeout; a step that times out or is killed ends the whole call
# (no further GPU step after a hang), any other failure is reported and the next step still runs.
t=$1; shift
timeout -k 10 "$t" "$@"
rc=$?
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[gpu_step] '$*' timed out / was killed (rc $rc): stopping this call" >&2; exit $rc; fi
echo "[gpu_step] '$1 $2 $3' rc=$rc" >&2
exit 0
