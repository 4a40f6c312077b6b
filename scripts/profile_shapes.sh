#!/bin/bash
# scripts/eval_shape_timing.py on a freshly generated 192-task set, then a PMC pass over case A and case B alone
# (VALU instructions per wave of the iterate / search kernels with and without particle contacts).  GPU box, repo root.
ROOT=$(pwd); TAG=${1:-r06}
OUT=$ROOT/gpurun_out/prof_shapes; SUM=$ROOT/gpurun_out/eval192_summary
rm -rf $OUT; mkdir -p $OUT $SUM
SET=$OUT/tasks.npz
python3 scripts/make_task_set.py 192 192 1 $SET | tail -1
python3 scripts/eval_shape_timing.py $SET 132 20 2>&1 | grep -v "TaskLoader\|amdgpu.ids" | tee $SUM/${TAG}_eval_shapes.txt
for CASE in A B; do
  cd /tmp && export TMPDIR=/tmp
  SHAPE_ONLY=$CASE rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/pmc$CASE -o s -- python3 $ROOT/scripts/eval_shape_timing.py $SET 132 4 > $OUT/pmc$CASE.log 2>&1
  cd $ROOT
  python3 - $OUT/pmc$CASE $CASE <<'PY' | tee -a $SUM/${TAG}_eval_shapes.txt
import sqlite3, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
con = sqlite3.connect(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for kname, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
    k = kname.split('(')[0][:44]
    acc[k][cname] += val
    if cname == 'SQ_WAVES': n[k] += 1
print("# PMC pass, case %s (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES): means over all launches" % sys.argv[2])
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
    if n[k] < 20: continue
    w = max(d.get('SQ_WAVES', 1), 1)
    print('%-46s launches %6d  waves/launch %7.0f  VALU/wave %6.0f  valu_active/wave_cycles %.3f  busy_cycles/launch %.0f' % (
        k, n[k], w / n[k], d.get('SQ_INSTS_VALU', 0) / w, d.get('SQ_ACTIVE_INST_VALU', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1), d.get('SQ_BUSY_CYCLES', 0) / n[k]))
PY
  find $OUT/pmc$CASE -name "*.db" -delete
done
rm -f $SET
