#!/bin/bash
# usage: tools/ktrace.sh <tag> <N> <D> [iters]   kernel-trace stats only (TVAE_PROF_DTYPE=bf16 for the bf16 path)
set -e
TAG=$1; N=$2; D=$3; IT=${4:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/prof_mmd.py $N $D $IT > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-90s %4s %12.1f us %6s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
