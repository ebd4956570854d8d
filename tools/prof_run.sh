#!/bin/bash
# usage: tools/prof_run.sh <tag> <N> <D> [iters]   (run on the GPU box via gpurun; writes gpurun_out/<tag>/)
# TVAE_PROF_DTYPE=bf16 in the environment profiles the bf16 path.
set -e
TAG=$1; N=$2; D=$3; IT=${4:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/prof_mmd.py
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $S $N $D $IT > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $S $N $D 2 > $OUT/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_lds -- python3 $S $N $D 2 > $OUT/pmc_lds.log 2>&1
find $OUT -name "*.csv" | head -20
