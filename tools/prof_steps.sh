#!/bin/bash
# usage: tools/prof_steps.sh <tag> <cfg> [<cfg> ...]  (on the GPU box): rocprofv3 kernel trace of a few eager steps per config
# -> gpurun_out/<tag>_<cfg>/  and a per-kernel table in gpurun_out/<tag>_<cfg>.txt
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for CFG in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_$CFG
  rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py $CFG 3 > $OUT.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kernel_table.py $OUT 3 > $OUT.txt
done
