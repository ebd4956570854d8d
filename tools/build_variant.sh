#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   -> transformer_vae_amd/variants/libtvae_<name>.so
# (A/B builds of single constants; the GPU tools load one with TVAE_LIB_OVERRIDE)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../transformer_vae_amd/csrc"
OUT=../variants; mkdir -p $OUT/obj_$NAME
for f in capi gemm rowops mmd mmd_bf16 mmd_pipe mmd_q32 mmd_q32s gram16 mmd_narrow mmd_n16 mmd_wide aux small_step; do
  # only the files the flags can touch are recompiled; the rest are the product objects
  if grep -q "TVAE_" <<< "$*" && grep -qE "$(echo "$*" | grep -oE 'TVAE_[A-Z_0-9]+' | paste -sd'|')" $f.hip; then
    EXTRA=""; [ $f = mmd_pipe ] && EXTRA="-fno-slp-vectorize"; [ $f = mmd_q32 ] && EXTRA="-fno-slp-vectorize -mllvm -pragma-unroll-threshold=65536"; [ $f = mmd_q32s ] && EXTRA="-fno-slp-vectorize -mllvm -pragma-unroll-threshold=65536 -mllvm -amdgpu-mfma-vgpr-form"      # as the Makefile
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EXTRA "$@" -c $f.hip -o $OUT/obj_$NAME/$f.o &
  else
    cp $f.o $OUT/obj_$NAME/$f.o
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/obj_$NAME/*.o -o $OUT/libtvae_$NAME.so
rm -rf $OUT/obj_$NAME
ls -la $OUT/libtvae_$NAME.so
