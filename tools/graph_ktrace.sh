#!/bin/bash
# usage: tools/graph_ktrace.sh <tag> <cfg>   kernel-trace of graph replays; prints per-kernel average duration
set -e
TAG=$1; CFG=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/small_graph_prof.py $CFG > $OUT.log 2>&1
tail -1 $OUT.log
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-80s %5s calls %8.2f us avg %6s%%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
