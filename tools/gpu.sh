#!/bin/bash
# Runs HERE (the build container): `tools/gpu.sh <timeout> '<command>'` = gpurun with retries while every GPU slot of the pod is busy
# (exit code 3: nothing charged).  One call at a time.
T="$1"; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
