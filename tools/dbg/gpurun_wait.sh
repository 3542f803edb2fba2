#!/bin/bash
# gpurun with patience: when no GPU slot is free (exit code 3: nothing ran, nothing was charged) wait and ask again, up
# to 12 times.  Any other outcome -- the command ran, was refused, failed -- is returned as it is (never re-run).
#   bash tools/dbg/gpurun_wait.sh --timeout 300 -- '<command>'
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 150
done
exit 3
