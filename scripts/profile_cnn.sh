#!/bin/bash
# rocprofv3 kernel stats + MFMA counters of the value network forward (scripts/cnn_timing.py).  GPU box, repo root.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_cnn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o cnn -- python3 $ROOT/scripts/cnn_timing.py > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $OUT/pmc -o cnn -- python3 $ROOT/scripts/cnn_timing.py > $OUT/pmc.log 2>&1
cd $ROOT
tail -4 $OUT/stats.log
python3 scripts/pmc_kernels.py $OUT/pmc > $OUT/pmc_kernels.txt 2>&1
cat $OUT/pmc_kernels.txt
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -r head -12
