#!/bin/bash
# usage: tools/sweep_splits.sh N D mode "cs1 cs2 ..."   -- whole-call time against the number of column splits (debug-env build)
N=$1; D=$2; MODE=$3; shift 3
for cs in $1; do
  echo "col_splits=$cs"
  TVAE_DEBUG_COL_SPLITS=$cs TVAE_LIB_OVERRIDE=transformer_vae_amd/variants/libtvae_dbg.so timeout -k 10 200 python3 -u tools/mmd_modes.py $N $D 2 $MODE | cut -c1-120 || exit 1
done
