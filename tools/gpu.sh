#!/bin/bash
# usage: tools/gpu.sh TIMEOUT 'command'   -- retries ONLY while gpurun reports "no slot free" (exit 3)
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[gpu.sh] no slot free (attempt $i), sleeping 120 s"
  sleep 120
done
exit 3
