#!/bin/bash
# usage: tools/ab_trace.sh <variant.so> N D   -- rocprofv3 kernel-trace averages of the MMD call, product vs variant library
V=$1; N=$2; D=$3
cd /tmp && export TMPDIR=/tmp
for tag in base var; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/abt_$tag; rm -rf $OUT
  if [ $tag = var ]; then export TVAE_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$V; fi
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_mmd.py $N $D 6 > $OUT.log 2>&1 || exit 1
  echo "== $tag"; python3 $GRAFT_REPO_ROOT/tools/kernel_table.py $OUT 6 | grep -E "mmd_|TOTAL" | cut -c1-70,100-170
done
