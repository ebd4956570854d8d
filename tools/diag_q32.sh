# A/B of the variant builds named on the command line ("product" = the shipped library): kernel times at the sizes in SIZES (default 32768)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" != product ]; then export TVAE_LIB_OVERRIDE=$GRAFT_REPO_ROOT/transformer_vae_amd/variants/libtvae_$v.so; else unset TVAE_LIB_OVERRIDE; fi
  echo "== $v"
  for n in ${SIZES:-32768}; do bash tools/ktrace.sh r3_kt_ab_${v}_$n $n ${DIM:-768} 2 | head -3 | cut -c40-150; done
done
