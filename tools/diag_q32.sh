# A/B of the variant builds named on the command line ("product" = the shipped library): kernel times at N = 32 768 (and cfg4 size with FULL=1)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" != product ]; then export TVAE_LIB_OVERRIDE=$GRAFT_REPO_ROOT/transformer_vae_amd/variants/libtvae_$v.so; else unset TVAE_LIB_OVERRIDE; fi
  echo "== $v"
  bash tools/ktrace.sh r3_kt_ab_$v 32768 768 3 | head -2 | cut -c40-150
  if [ -n "$FULL" ]; then bash tools/ktrace.sh r3_kt_ab4_$v 245760 768 1 | head -2 | cut -c40-150; fi
done
