#!/bin/bash
# The evaluation loop at the shape bench.py quotes as eval_loop.continuous (384 generated hard tasks through 192 slots, 3 actions):
#   1. the task set once (scripts/make_task_set.py), outside every profiler
#   2. the loop unprofiled (wall clock, frames, mean active episodes) and with contact-candidate sampling
#   3. rocprofv3 --kernel-trace --stats of the loop alone -> per-kernel totals, time per frame by kernel, device-idle fraction
#   4. rocprofv3 --pmc on a 192-task / 192-slot / 1-action run of the same set -> VALU per wave, waves, VALU-active per kernel
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
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $OUT/pmc -o ev -- python3 $ROOT/scripts/eval_from_set.py $SET $S 1 > $OUT/pmc.log 2>&1
cd $ROOT
python3 - $OUT/pmc <<'PY' | tee $SUM/${TAG}_eval192_pmc.txt
import sqlite3, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
con = sqlite3.connect(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
for kname, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
    k = kname.split('(')[0][:44]
    acc[k][cname] += val
    if cname == 'SQ_WAVES': n[k] += 1
print("# rocprofv3 --pmc pass of scripts/eval_from_set.py <384-task set> 192 1 (192 slots, one action per episode): per-kernel means over ALL its launches")
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
    if n[k] < 100: continue
    w = max(d.get('SQ_WAVES', 1), 1)
    print('%-46s launches %7d  waves/launch %7.0f  VALU/wave %6.0f  SALU/wave %5.0f  VMEM_RD/wave %6.1f  valu_active/wave_cycles %.3f  busy_cycles/launch %.0f' % (
        k, n[k], w / n[k], d.get('SQ_INSTS_VALU', 0) / w, d.get('SQ_INSTS_SALU', 0) / w, d.get('SQ_INSTS_VMEM_RD', 0) / w,
        d.get('SQ_ACTIVE_INST_VALU', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1), d.get('SQ_BUSY_CYCLES', 0) / n[k]))
PY
find $OUT -name "*.db" -delete; rm -f $SET
