#!/bin/bash
# rocprofv3 kernel stats of BASELINE.json configs[1]: render (720 x 720 RGBA + depth) and the device observation stage of one
# crumpled 64x64 episode with the two pickers (scripts/render_timing.py: 21 renders + 21 render+observe calls).  GPU box, repo root.
ROOT=$(pwd); TAG=${1:-r04}
OUT=$ROOT/gpurun_out/prof_render
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/render_summary
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o rd -- python3 $ROOT/scripts/render_timing.py > $OUT/stats.log 2>&1
cd $ROOT
cat $OUT/stats.log | grep "ms per call" | tee $ROOT/gpurun_out/render_summary/${TAG}_render_timing.txt
python3 - $OUT/stats $ROOT/gpurun_out/render_summary/${TAG}_render_kernel_stats.csv <<'PY'
import csv, os, sqlite3, sys
db = [os.path.join(r, f) for r, _, fs in os.walk(sys.argv[1]) for f in fs if f.endswith(".db")][0]
con = sqlite3.connect(db)
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel (scripts/render_timing.py: 80 solver frames, then 21 x fs_render and 21 x render + fs_observe of one 64x64 episode)", "calls", "total_us", "average_us", "percent_of_gpu_time"])
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if "raster" in name or "shade" in name or "sphere" in name or "normal" in name or "observe" in name or "obs" in name or "label" in name or pct >= 1.0:
            w.writerow([name[:120], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.2f}"])
PY
cat $ROOT/gpurun_out/render_summary/${TAG}_render_kernel_stats.csv | cut -c1-160
find $OUT -name "*.db" -delete
