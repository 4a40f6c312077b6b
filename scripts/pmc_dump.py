"""Development helper: per-launch averages of rocprofv3 PMC counters (rocpd sqlite) for the LAST n launches of a kernel.
    python scripts/pmc_dump.py <dir> [kernel-substring] [n]"""
import os, sqlite3, sys
src = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "fs_k_fused_step"; last = int(sys.argv[3]) if len(sys.argv) > 3 else 20
path = None
for root, _, files in os.walk(src):
    for f in files:
        if f.endswith(".db"):
            path = os.path.join(root, f)
con = sqlite3.connect(path)
cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
order = "start" if "start" in cols else ("dispatch_id" if "dispatch_id" in cols else "rowid")
names = [r[0] for r in con.execute("select distinct counter_name from counters_collection where kernel_name like ?", (f"%{kern}%",))]
for nm in names:
    rows = list(con.execute(f"select value, duration from counters_collection where kernel_name like ? and counter_name = ? order by {order}", (f"%{kern}%", nm)))
    # a counter may be reported per XCD / SE: sum rows that share a dispatch
    if "dispatch_id" in cols:
        rows = list(con.execute(f"select sum(value), max(duration) from counters_collection where kernel_name like ? and counter_name = ? group by dispatch_id order by dispatch_id", (f"%{kern}%", nm)))
    rows = rows[-last:]
    print("%-24s %16.1f per launch  (%d launches, %.1f us)" % (nm, sum(r[0] for r in rows) / len(rows), len(rows), sum(r[1] for r in rows) / len(rows) / 1e3))
