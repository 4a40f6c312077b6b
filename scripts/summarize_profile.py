"""Turns the rocprofv3 (rocpd sqlite) outputs of scripts/profile_bench.sh into the small summaries committed under
profiles/:  <tag>_kernel_stats.csv (per-kernel calls / total / average, = `--stats`), <tag>_pmc.json (per-launch
averages of the PMC counters for the dominant kernel) and hbm_traffic.json (HBM bytes per launch read by bench.py).

    python scripts/summarize_profile.py gpurun_out/prof_r01 r01 --episodes 256
"""
import argparse
import csv
import json
import os
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("tag")
ap.add_argument("--episodes", type=int, default=256)
ap.add_argument("--kernel", default="fs_k_fused_grid64")
ap.add_argument("--last", type=int, default=30, help="average over the last N launches of the kernel = the timed region "
                                                     "of `bench.py --steps N` (pre-roll and warm-up launches excluded)")
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)


def db(sub):
    for f in os.listdir(os.path.join(args.src, sub)):
        if f.endswith(".db"):
            return sqlite3.connect(os.path.join(args.src, sub, f))
    raise FileNotFoundError(sub)


con = db("stats")
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(os.path.join(OUT, f"{args.tag}_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
dom = [r for r in rows if args.kernel in r[0]][0]
last = [r[0] for r in con.execute("select duration from kernels where name like ? order by start desc limit ?",
                                  (f"%{args.kernel}%", args.last))]
summary = {"tag": args.tag,
           "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline "
                      "--no-secondary --no-parity",
           "kernel": dom[0], "calls": dom[1], "average_us_all_launches": dom[3], "percent_of_gpu_time": dom[4],
           "episodes": args.episodes, "waves_per_simd": 4,
           "timed_launches": len(last), "average_us": sum(last) / len(last) / 1e3, "counters": {}}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mem"):
    try:
        c = db(sub)
    except FileNotFoundError:
        continue
    names = [r[0] for r in c.execute("select distinct counter_name from counters_collection where kernel_name like ?",
                                     (f"%{args.kernel}%",))]
    for name in names:  # the last N launches = the timed region
        q = ("select value, duration from counters_collection where kernel_name like ? and counter_name = ? "
             "order by start desc limit ?")
        vals = list(c.execute(q, (f"%{args.kernel}%", name, args.last)))
        summary["counters"][name] = {"avg_per_launch": sum(v for v, _ in vals) / len(vals), "launches": len(vals),
                                     "avg_duration_ns": sum(d for _, d in vals) / len(vals)}
cn = summary["counters"]
if "FETCH_SIZE" in cn and "WRITE_SIZE" in cn:
    # FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X guide (HBM section): on gfx950 FETCH_SIZE reports half of the bytes of
    # wide coalesced streaming reads -> double it; WRITE_SIZE is used as reported (uncalibrated).
    fetch = cn["FETCH_SIZE"]["avg_per_launch"] * 1024.0
    write = cn["WRITE_SIZE"]["avg_per_launch"] * 1024.0
    summary["hbm_bytes_per_launch"] = {"fetch_reported": fetch, "fetch_corrected_x2": 2 * fetch, "write": write,
                                       "total_corrected": 2 * fetch + write}
    with open(os.path.join(OUT, "hbm_traffic.json"), "w") as fh:
        json.dump({"tag": args.tag, "episodes": args.episodes, "bytes_per_launch": 2 * fetch + write,
                   "note": "rocprofv3 FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, KiB -> bytes, averaged over the "
                           "launches of " + dom[0]}, fh, indent=1)
if "SQ_ACTIVE_INST_VALU" in cn and "SQ_WAVE_CYCLES" in cn:
    summary["valu_active_fraction_of_wave_cycles"] = (cn["SQ_ACTIVE_INST_VALU"]["avg_per_launch"] /
                                                      cn["SQ_WAVE_CYCLES"]["avg_per_launch"])
with open(os.path.join(OUT, f"{args.tag}_pmc.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps(summary, indent=1))
