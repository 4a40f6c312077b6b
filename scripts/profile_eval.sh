#!/bin/bash
# rocprofv3 kernel stats of the evaluation loop (scripts/eval_loop_demo.py 32 3: BASELINE.json configs[4] in miniature).
# GPU box, repo root.  Writes gpurun_out/eval_summary/r02_eval_kernel_stats.csv (copy it to profiles/).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_eval
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/eval_summary
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o ev -- python3 $ROOT/scripts/eval_loop_demo.py 32 3 > $OUT/stats.log 2>&1
cd $ROOT
tail -2 $OUT/stats.log | tee $ROOT/gpurun_out/eval_summary/r02_eval_run.txt
python3 - $OUT/stats $ROOT/gpurun_out/eval_summary/r02_eval_kernel_stats.csv <<'PY'
import csv, os, sqlite3, sys
db = [os.path.join(r, f) for r, _, fs in os.walk(sys.argv[1]) for f in fs if f.endswith(".db")][0]
con = sqlite3.connect(db)
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel (scripts/eval_loop_demo.py 32 3: task generation + 32 episodes x 3 actions)", "calls", "total_us", "average_us", "percent_of_gpu_time"])
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if pct >= 0.05:
            w.writerow([name[:120], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.2f}"])
PY
cat $ROOT/gpurun_out/eval_summary/r02_eval_kernel_stats.csv | cut -c1-150
find $OUT -name "*.db" -delete
