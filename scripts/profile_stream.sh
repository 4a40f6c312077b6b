#!/bin/bash
# rocprofv3 kernel stats of the streaming back-end on 64 episodes of a 104x104 cloth -> profiles/r02_stream_kernel_stats.csv
# (run on the GPU box from the repo root; copy gpurun_out/stream_summary/* to profiles/).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_stream
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/stream_summary
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o st -- python3 $ROOT/scripts/large_cloth_timing.py 104 64 > $OUT/stats.log 2>&1
cd $ROOT
tail -1 $OUT/stats.log
python3 - $OUT/stats $ROOT/gpurun_out/stream_summary/r02_stream_kernel_stats.csv <<'PY'
import csv, os, sqlite3, sys
db = [os.path.join(r, f) for r, _, fs in os.walk(sys.argv[1]) for f in fs if f.endswith(".db")][0]
con = sqlite3.connect(db)
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel (rocprofv3 --kernel-trace --stats -- python3 scripts/large_cloth_timing.py 104 64: streaming back-end, 64 episodes of a 104x104 cloth, 13 steps)", "calls", "total_us", "average_us", "percent"])
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if pct >= 0.5:
            w.writerow([name[:120], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.2f}"])
PY
cat $ROOT/gpurun_out/stream_summary/r02_stream_kernel_stats.csv | cut -c1-140
find $OUT -name "*.db" -delete
