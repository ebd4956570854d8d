#!/bin/bash
# usage: gpurun_wait.sh <outfile> <timeout> <command>  -- waits for a free box (exit 3 = none free, nothing ran), then runs once
OUT=$1; TMO=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $OUT 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" $OUT; then exit $rc; fi
  sleep 75
done
exit 3
