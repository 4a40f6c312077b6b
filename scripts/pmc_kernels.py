"""Development helper: rocprofv3 PMC counters (rocpd sqlite) summed per kernel name over a whole run.
    python scripts/pmc_kernels.py <dir>"""
import os, sqlite3, sys
path = None
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith(".db"):
            path = os.path.join(root, f)
con = sqlite3.connect(path)
cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
disp = "dispatch_id" if "dispatch_id" in cols else "rowid"
per = {}
for kern, name, value, dur, n in con.execute(
        f"select kernel_name, counter_name, sum(value), sum(d), count(*) from (select kernel_name, counter_name, sum(value) as value, "
        f"max(duration) as d from counters_collection group by {disp}, counter_name) group by kernel_name, counter_name"):
    k = per.setdefault(kern, {"_ns": 0.0, "_n": 0})
    k[name] = value; k["_ns"] = dur; k["_n"] = n
tot = {}
for kern, c in sorted(per.items(), key=lambda kv: -kv[1]["_ns"]):
    print("%-90s launches %5d  %9.1f us total  " % (kern[:90], c["_n"], c["_ns"] / 1e3) +
          "  ".join("%s=%.4g" % (k, v) for k, v in sorted(c.items()) if not k.startswith("_")))
    for k, v in c.items():
        tot[k] = tot.get(k, 0.0) + v
print("TOTAL " + "  ".join("%s=%.6g" % (k, v) for k, v in sorted(tot.items())))
if "SQ_INSTS_VALU_MFMA_MOPS_F32" in tot and tot.get("_ns"):
    flops = tot["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
    print("fp32 MFMA flops %.4g over %.3f ms of kernel time = %.2f TFLOP/s" % (flops, tot["_ns"] / 1e6, flops / tot["_ns"] / 1e3))
if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and tot.get("SQ_BUSY_CYCLES"):
    print("MFMA busy cycles / SQ busy cycles = %.4f" % (tot["SQ_VALU_MFMA_BUSY_CYCLES"] / tot["SQ_BUSY_CYCLES"]))
