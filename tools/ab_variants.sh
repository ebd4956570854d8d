#!/bin/bash
# usage (GPU box): tools/ab_variants.sh "<cfg> [<cfg> ...]" <variant> [<variant> ...]
#   per-kernel tables (rocprofv3 kernel trace of eager steps) for the product library and each
#   transformer_vae_amd/variants/libtvae_<variant>.so; prints the lines matching $AB_GREP (default: gemm|TOTAL)
CFGS=$1; shift
PAT=${AB_GREP:-gemm|TOTAL}
show() { for c in $CFGS; do echo "== $1 $c"; grep -E "$PAT" $GRAFT_REPO_ROOT/gpurun_out/$1_$c.txt | sed 's/void tvae::(anonymous namespace):://' | cut -c1-64,100-160; done; }
tools/prof_steps.sh ab_base $CFGS || exit 1
cd $GRAFT_REPO_ROOT; show ab_base
for V in "$@"; do
  cd $GRAFT_REPO_ROOT
  TVAE_LIB_OVERRIDE=$GRAFT_REPO_ROOT/transformer_vae_amd/variants/libtvae_$V.so tools/prof_steps.sh ab_$V $CFGS || exit 1
  cd $GRAFT_REPO_ROOT; show ab_$V
done
