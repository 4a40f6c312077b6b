#!/bin/bash
# rocprofv3 kernel stats of the row-f workloads (fling primitive, prepare_image, action selection).  GPU box, repo root.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_rows
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/fling -o fling -- python3 $ROOT/scripts/fling_timing.py 64 > $OUT/fling.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prep -o prep -- python3 $ROOT/scripts/prepare_image_timing.py > $OUT/prep.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/action -o action -- python3 $ROOT/tests/soak/action_timing.py > $OUT/action.log 2>&1
cd $ROOT
tail -1 $OUT/fling.log; tail -1 $OUT/prep.log; tail -1 $OUT/action.log
