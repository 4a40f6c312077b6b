ROOT=$(pwd)
rm -rf $ROOT/gpurun_out/pmc_ff
cd /tmp && export TMPDIR=/tmp
export FLINGSIM_STREAM_GROUPS=1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $ROOT/gpurun_out/pmc_ff -o ff -- python3 $ROOT/scripts/freefall64.py > /dev/null 2>&1
cd $ROOT
python3 - <<'PY'
import sqlite3, glob, collections
f = glob.glob('gpurun_out/pmc_ff/**/*.db', recursive=True)[0]
con = sqlite3.connect(f)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for kname, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
    acc[kname.split('(')[0][:40]][cname].append(val)
for k, d in acc.items():
    n = len(next(iter(d.values())))
    if n < 100: continue
    m = {c: sum(v[-200:]) / len(v[-200:]) for c, v in d.items()}
    w = max(m.get('SQ_WAVES', 1), 1)
    print(k, 'launches', n, ' waves %.0f  VALU/wave %.0f  valu_active/wave_cycles %.3f' % (w, m.get('SQ_INSTS_VALU', 0) / w, m.get('SQ_ACTIVE_INST_VALU', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1)))
PY
find $ROOT/gpurun_out/pmc_ff -name "*.db" -delete
