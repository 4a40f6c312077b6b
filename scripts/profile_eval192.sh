#!/bin/bash
# The evaluation loop at the shape bench.py quotes as eval_loop.continuous (384 generated hard tasks through 192 slots, 3 actions):
#   1. the task set once (scripts/make_task_set.py), outside every profiler
#   2. the loop unprofiled (wall clock, frames, mean active episodes) and with contact-candidate sampling
#   3. rocprofv3 --kernel-trace --stats of the loop alone -> per-kernel totals, time per frame by kernel, device-idle fraction
#   (PMC counters of this launch shape: scripts/profile_shapes.sh)
# GPU box, repo root.  Summaries: gpurun_out/eval192_summary/${TAG}_eval192_* (copy to profiles/).
ROOT=$(pwd); TAG=${1:-r06}; N=${2:-384}; S=${3:-192}
SUM=$ROOT/gpurun_out/eval192_summary; OUT=$ROOT/gpurun_out/prof_eval192
rm -rf $OUT; mkdir -p $OUT $SUM
SET=$OUT/tasks.npz
python3 scripts/make_task_set.py $N $S 1 $SET 2>&1 | tail -1 | tee $SUM/${TAG}_eval192_run.txt
python3 scripts/eval_from_set.py $SET $S 3 2>&1 | grep -v TaskLoader | tee -a $SUM/${TAG}_eval192_run.txt
EVAL_CONTACTS=40 python3 scripts/eval_from_set.py $SET $S 3 2>&1 | grep "contact candidates" | tee -a $SUM/${TAG}_eval192_run.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o ev -- python3 $ROOT/scripts/eval_from_set.py $SET $S 3 > $OUT/stats.log 2>&1
cd $ROOT
grep -A2 "^eval loop" $OUT/stats.log | sed 's/^/under rocprofv3 --kernel-trace: /' | tee -a $SUM/${TAG}_eval192_run.txt
python3 scripts/summarize_eval_trace.py $OUT/stats $SUM/${TAG}_eval192_kernel_stats.csv | tee -a $SUM/${TAG}_eval192_run.txt
find $OUT/stats -name "*.db" -delete
# (A PMC pass over the loop ITSELF is impractical: counter collection serialises its ~2 M dispatches -- the first attempt produced a
#  database the box could not even summarise, a second one restricted to a 5 s --collection-period window did not finish in 20 minutes.
#  The counters of this launch shape are taken on the loop-less reconstruction instead: scripts/profile_shapes.sh, case A / B.)
find $OUT -name "*.db" -delete; rm -f $SET
