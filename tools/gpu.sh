#!/bin/bash
# usage: tools/gpu.sh <timeout_s> <logfile> <command...>   -- gpurun, retried only while the pool has no free slot (exit code 3)
T=$1; LOG=$2; shift 2
for attempt in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
