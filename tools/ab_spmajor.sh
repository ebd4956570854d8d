#!/bin/bash
# usage (GPU box): tools/ab_spmajor.sh N D "<cs|0> <sp_major|-> [nosync]" ...   -- time + L2->fabric read bytes of the MMD call
# under workgroup-order / column-split overrides.  Variant libraries (built here, before the GPU call):
#   tools/build_variant.sh dbgenv -DTVAE_DEBUG_ENV        (env overrides TVAE_DEBUG_COL_SPLITS / _SP_MAJOR / _NO_SYNC)
#   AB_LIB=<name> picks another transformer_vae_amd/variants/libtvae_<name>.so (e.g. one built from an older commit)
N=$1; D=$2; shift 2
export TVAE_LIB_OVERRIDE=$GRAFT_REPO_ROOT/transformer_vae_amd/variants/libtvae_${AB_LIB:-dbgenv}.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  set -- $v
  export TVAE_DEBUG_COL_SPLITS=$1 TVAE_DEBUG_SP_MAJOR=$2
  [ "$1" = "0" ] && unset TVAE_DEBUG_COL_SPLITS
  [ "$2" = "-" ] && unset TVAE_DEBUG_SP_MAJOR
  unset TVAE_DEBUG_NO_SYNC; [ "$3" = "nosync" ] && export TVAE_DEBUG_NO_SYNC=1
  t=$(timeout -k 10 120 python3 $GRAFT_REPO_ROOT/tools/prof_mmd.py $N $D 4 2>/dev/null | grep "^last") || exit 1
  OUT=$GRAFT_REPO_ROOT/gpurun_out/spm_$1_$2_$3
  rm -rf $OUT
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_mmd.py $N $D 2 > $OUT.log 2>&1 || exit 1
  f=$(python3 - $OUT <<'PY'
import csv, glob, sys
tot = 0.0; n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mmd_tile_kernel" in r["Kernel_Name"] and "true" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print("grad-kernel FETCH_SIZE per launch: %.1f GB (x2 gfx950 correction: %.1f GB), %d launches" % (tot / n / 1e6, 2 * tot / n / 1e6, n))
PY
)
  echo "cs=$1 sp_major=$2 $3 | $t | $f"
done
