#!/bin/bash
# gpurun with patience: retries only while NO SLOT was free (exit code 3: nothing ran, nothing was charged); any other outcome is final.
#   tools/gpurun_retry.sh <timeout seconds> '<command>'
T="$1"; shift
for attempt in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
