"""profiles/r01_rows_kernel_stats.csv from the rocpd databases written by scripts/profile_rows.sh."""
import csv, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_rows")
rows = []
for sub in ("fling", "prep", "action"):
    d = os.path.join(src, sub)
    db = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")][0]
    con = sqlite3.connect(db)
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if pct >= 0.5:
            rows.append([sub, name[:110], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.2f}"])
with open(os.path.join(ROOT, "profiles", "r01_rows_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["workload (scripts/profile_rows.sh)", "kernel", "calls", "total_us", "average_us", "percent_of_gpu_time"])
    w.writerows(rows)
print(open(os.path.join(ROOT, "profiles", "r01_rows_kernel_stats.csv")).read())
