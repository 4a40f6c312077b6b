#!/bin/bash
# rocprofv3 kernel stats of the streaming back-end on BASELINE.json configs[2]: 64 crumpled 64x64 episodes in one launch
# sequence (tests/soak/ab_fused.py 64 1).  GPU box, repo root; writes gpurun_out/stream_summary/${TAG}_stream64_kernel_stats.csv.
ROOT=$(pwd); TAG=${1:-r04}
mkdir -p $ROOT/gpurun_out/stream_summary
rm -rf $ROOT/gpurun_out/prof_s64
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_s64 -o s64 -- python3 $ROOT/tests/soak/ab_fused.py 64 1 > /dev/null 2>&1
cd $ROOT
TAG=$TAG python3 - <<'PY'
import csv, sqlite3, glob, statistics, os
f = glob.glob('gpurun_out/prof_s64/**/*.db', recursive=True)[0]
con = sqlite3.connect(f)
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open('gpurun_out/stream_summary/' + os.environ.get('TAG', 'r04') + '_stream64_kernel_stats.csv', 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(["kernel (rocprofv3 --kernel-trace --stats -- python3 tests/soak/ab_fused.py 64 1: streaming back-end, 64 crumpled 64x64 episodes)", "calls", "total_us", "average_us", "percent"])
    for name, calls, total, avg, pct in rows:
        if pct >= 0.01:
            w.writerow([name[:120], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.2f}"])
        if pct >= 0.5: print(name[:90], calls, round(avg, 3), round(pct, 2))
ks = list(con.execute("select start,end from kernels order by start"))
print('launches', len(ks), 'median duration ns', statistics.median(e - s for s, e in ks),
      'median gap ns', statistics.median(ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)))
PY
find $ROOT/gpurun_out/prof_s64 -name "*.db" -delete
