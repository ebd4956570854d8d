#!/bin/bash
# usage: tools/prof_pmc.sh <tag> <N> <D> "<counters>" [more counter sets...]   one rocprofv3 --pmc pass per set
set -e
TAG=$1; N=$2; D=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/prof_mmd.py
i=0
for set in "$@"; do
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $S $N $D 2 > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, collections, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "mmd_" in k: print(k, {a: "%.4g" % b for a, b in v.items()})
PY
