cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_s64 -o s64 -- python3 $GRAFT_REPO_ROOT/tests/soak/ab_fused.py 64 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sqlite3, glob
f=glob.glob('gpurun_out/prof_s64/**/*.db', recursive=True)[0]
con=sqlite3.connect(f)
for r in con.execute("select name,total_calls,average,percentage from top_kernels limit 12"): print(r)
# gaps: kernel dispatch start/end
try:
    rows=list(con.execute("select start,end from kernels order by start"))
    import statistics
    durs=[e-s for s,e in rows]; gaps=[rows[i+1][0]-rows[i][1] for i in range(len(rows)-1)]
    print('n',len(rows),'median dur',statistics.median(durs),'median gap',statistics.median(gaps))
except Exception as ex: print('gap query failed',ex); print([r for r in con.execute("select name from sqlite_master where type='table' or type='view'")][:40])
PY
