"""Development helper: rocprofv3 --kernel-trace --stats database of an evaluation-loop run -> per-kernel CSV + where a frame's
time goes: kernel time by family per frame (a frame = 120 iterate launches per launch chain), the device-busy union of all
dispatches, and the idle gaps between them.   usage: summarize_eval_trace.py <dir> <out.csv>"""
import csv, os, sqlite3, sys
import numpy as np

db = [os.path.join(r, f) for r, _, fs in os.walk(sys.argv[1]) for f in fs if f.endswith(".db")][0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel (scripts/eval_from_set.py: evaluation loop alone, stored task set)", "calls", "total_ns", "average_ns", "percent_of_kernel_time"])
    for name, calls, total, avg, pct in rows:
        if pct >= 0.03:
            w.writerow([name[:120], calls, f"{total:.0f}", f"{avg:.1f}", f"{pct:.2f}"])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
ks = np.array(list(con.execute("select start, end from kernels order by start")), dtype=np.int64)
names = [r[0] for r in con.execute("select name from kernels order by start")]
dur = ks[:, 1] - ks[:, 0]
span = ks[:, 1].max() - ks[:, 0].min()
# union of busy intervals
end_so_far = np.maximum.accumulate(ks[:, 1])
gap = np.maximum(ks[1:, 0] - end_so_far[:-1], 0)
busy = span - gap.sum()
fam = {}
for nm, d in zip(names, dur):
    k = nm.split("(")[0].replace("void ", "")
    k = k.split("<")[0]
    a = fam.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(d)
iters = fam.get("fs_k_iterate_gridl", [0, 0])[0] + fam.get("fs_k_iterate_grid", [0, 0])[0]
print("kernel trace: %d dispatches over %.2f s; sum of kernel durations %.2f s; device busy (union) %.2f s = %.1f %% of the span; "
      "idle %.2f s in %d gaps (median %.1f us, %.1f %% of the idle time in gaps > 100 us)" % (
          len(dur), span / 1e9, dur.sum() / 1e9, busy / 1e9, 100.0 * busy / span, gap.sum() / 1e9, int((gap > 0).sum()),
          float(np.median(gap[gap > 0])) / 1e3 if (gap > 0).any() else 0.0, 100.0 * gap[gap > 100000].sum() / max(gap.sum(), 1)))
print("overlap: sum of durations / busy union = %.2f (two concurrent launch chains + the service lane)" % (dur.sum() / busy))
if qcol:
    per_q = {}
    for (q,), d in zip(con.execute(f"select {qcol} from kernels order by start"), dur):
        a = per_q.setdefault(q, [0, 0]); a[0] += 1; a[1] += int(d)
    print("per %s: " % qcol + "; ".join("%s: %d dispatches %.2f s" % (q, a[0], a[1] / 1e9) for q, a in sorted(per_q.items(), key=lambda kv: -kv[1][1])))
print("kernel time by family (share of the sum of durations):")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-28s %9d launches  %8.3f s  %5.1f %%  mean %8.2f us" % (k, c, t / 1e9, 100.0 * t / dur.sum(), t / c / 1e3))
