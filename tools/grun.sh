#!/bin/bash
# developer helper: gpurun with retries while no GPU slot is free (exit code 3: nothing ran, nothing charged)
# usage: tools/grun.sh <tag> <timeout> '<command>'
TAG=$1; TMO=$2; CMD=$3
mkdir -p gpurun_out/$TAG
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "mkdir -p gpurun_out/$TAG && $CMD" > gpurun_out/$TAG/call.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
