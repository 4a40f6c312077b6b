#!/bin/bash
# rocprofv3 kernel stats of the evaluation loop (BASELINE.json configs[4]): scripts/eval_async_timing.py 64 32 3 = 64 generated
# hard tasks through 32 slots with evaluate.run_tasks.  GPU box, repo root.
# Writes gpurun_out/eval_summary/${TAG}_eval_kernel_stats.csv (copy it to profiles/).
ROOT=$(pwd); TAG=${1:-r04}
OUT=$ROOT/gpurun_out/prof_eval
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/eval_summary
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o ev -- python3 $ROOT/scripts/eval_async_timing.py 64 32 3 > $OUT/stats.log 2>&1
cd $ROOT
grep -A1 "async slots" $OUT/stats.log | tee $ROOT/gpurun_out/eval_summary/${TAG}_eval_run.txt
python3 - $OUT/stats $ROOT/gpurun_out/eval_summary/${TAG}_eval_kernel_stats.csv <<'PY'
import csv, os, sqlite3, sys
db = [os.path.join(r, f) for r, _, fs in os.walk(sys.argv[1]) for f in fs if f.endswith(".db")][0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("sum of kernel durations: %.3f s" % (sum(r[2] for r in rows) * 1e-9 if rows and rows[0][2] > 1e7 else sum(r[2] for r in rows) * 1e-6))
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel (scripts/eval_async_timing.py 64 32 3: task generation + 64 episodes x 3 actions over 32 slots)", "calls", "total_us", "average_us", "percent_of_gpu_time"])
    for name, calls, total, avg, pct in rows:
        if pct >= 0.05:
            w.writerow([name[:120], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.2f}"])
PY
cat $ROOT/gpurun_out/eval_summary/${TAG}_eval_kernel_stats.csv | cut -c1-150
find $OUT -name "*.db" -delete
