#!/bin/bash
# PMC counters of search-dominated launches (scripts/search_only.py).  Run on the GPU box from the repo root.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_search
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/a -o a -- python3 $ROOT/scripts/search_only.py 1 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/b -o b -- python3 $ROOT/scripts/search_only.py 1 > $OUT/b.log 2>&1
cd $ROOT
for t in a b; do python3 scripts/pmc_dump.py $OUT/$t; done
tail -3 $OUT/a.log $OUT/b.log
