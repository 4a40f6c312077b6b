#!/bin/bash
# PMC pass over the evaluation loop ITSELF at the shape the bench quotes (384 tasks / 192 slots): counters are collected for a
# 5 s window of the steady state only (rocprofv3 --collection-period; the whole run would be ~2 M dispatches x counters x XCDs,
# which round 6's first attempt could not even summarise).  GPU box, repo root -> gpurun_out/eval192_summary/${TAG}_eval192_pmc.txt
ROOT=$(pwd); TAG=${1:-r06}
OUT=/tmp/pmc_eval192; SUM=$ROOT/gpurun_out/eval192_summary
rm -rf $OUT; mkdir -p $OUT $SUM
SET=$OUT/tasks.npz
python3 scripts/make_task_set.py 384 192 1 $SET | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -P 35:5:1 -d $OUT/pmc -o ev -- python3 $ROOT/scripts/eval_from_set.py $SET 192 3 > $OUT/pmc.log 2>&1
cd $ROOT
grep "^chunk bounds\|eval loop" $OUT/pmc.log | head -2
python3 - $OUT/pmc <<'PY' | tee $SUM/${TAG}_eval192_pmc.txt
import sqlite3, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
con = sqlite3.connect(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for kname, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
    k = kname.split('(')[0][:44]
    acc[k][cname] += val
    if cname == 'SQ_WAVES': n[k] += 1
print("# rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -P 35:5:1 over scripts/eval_from_set.py <384-task set> 192 3:")
print("# the evaluation loop itself, a 5 s window of its steady state (counter collection serialises the kernels); means over the window's launches")
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
    if n[k] < 20: continue
    w = max(d.get('SQ_WAVES', 1), 1)
    print('%-46s launches %7d  waves/launch %7.0f  VALU/wave %6.0f  valu_active/wave_cycles %.3f  busy_cycles/launch %.0f' % (
        k, n[k], w / n[k], d.get('SQ_INSTS_VALU', 0) / w, d.get('SQ_ACTIVE_INST_VALU', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1), d.get('SQ_BUSY_CYCLES', 0) / n[k]))
PY
rm -rf $OUT
