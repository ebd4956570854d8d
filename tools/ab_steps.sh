#!/bin/bash
# usage: tools/ab_steps.sh <variant.so> <cfg> [<cfg> ...]   -- per-kernel tables of eager steps, product vs variant library
V=$1; shift
tools/prof_steps.sh abbase "$@" || exit 1
TVAE_LIB_OVERRIDE=$GRAFT_REPO_ROOT/$V tools/prof_steps.sh abvar "$@" || exit 1
