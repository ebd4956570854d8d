#!/bin/bash
# usage: tools/prof_bench.sh <tag> <bench args...>  (on the GPU box) -> gpurun_out/<tag>/{trace,pmc_fetch,pmc_write,pmc_sq,pmc_lds}
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B "$@" --no-cpu-baseline --no-side-configs > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B "$@" --no-cpu-baseline --no-side-configs > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B "$@" --no-cpu-baseline --no-side-configs > $OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $B "$@" --no-cpu-baseline --no-side-configs > $OUT/pmc_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_lds -- python3 $B "$@" --no-cpu-baseline --no-side-configs > $OUT/pmc_lds.log 2>&1
tail -1 $OUT/trace.log | cut -c1-200
