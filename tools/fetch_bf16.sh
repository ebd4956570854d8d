#!/bin/bash
# usage (GPU box): tools/fetch_bf16.sh N D [lib-variant]  -- time + L2->fabric read bytes of the bf16 MMD gradient launch
N=$1; D=$2
[ -n "$3" ] && export TVAE_LIB_OVERRIDE=$GRAFT_REPO_ROOT/transformer_vae_amd/variants/libtvae_$3.so
export TVAE_PROF_DTYPE=bf16
cd /tmp && export TMPDIR=/tmp
t=$(timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/prof_mmd.py $N $D 4 2>/dev/null | grep "^last") || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/fb16_${3:-prod}
rm -rf $OUT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_mmd.py $N $D 2 > $OUT.log 2>&1 || exit 1
python3 - $OUT "$t" <<'PY'
import csv, glob, sys
tot = 0.0; n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mmd_bf16_kernel" in r["Kernel_Name"] and "true" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print(sys.argv[2], "| bf16 grad-kernel FETCH_SIZE x2 per launch: %.1f GB, %d launches" % (2 * tot / max(n, 1) / 1e6, n))
PY
