#!/bin/bash
# usage: tools/ab.sh <variant.so> "N D" ["N D" ...]   -- prof_mmd timing, product library vs a variant build
V=$1; shift
for nd in "$@"; do
  set -- $nd
  a=$(timeout -k 10 120 python3 tools/prof_mmd.py $1 $2 5 2>/dev/null | grep "^last") || exit 1
  b=$(TVAE_LIB_OVERRIDE=$V timeout -k 10 120 python3 tools/prof_mmd.py $1 $2 5 2>/dev/null | grep "^last") || exit 1
  echo "N=$1 D=$2 | base: $a | variant: $b"
done
