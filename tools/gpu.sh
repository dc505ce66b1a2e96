#!/bin/bash
# Usage (build container): tools/gpu.sh <log> <timeout_s> '<command>'  -- gpurun with retries while every GPU slot of the pod is busy
# (exit code 3: nothing charged).  The call's tail goes to <log>; gpurun_out/.last_call.json has the verdict.
log=$1; to=$2; shift; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
    rc=$?
    [ $rc = 3 ] || exit $rc
    sleep 45
done
exit 3
