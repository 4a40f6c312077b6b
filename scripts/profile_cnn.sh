#!/bin/bash
# rocprofv3 kernel stats + MFMA counters of the value network forward (scripts/cnn_timing.py).  GPU box, repo root.
ROOT=$(pwd); TAG=${1:-r04}
OUT=$ROOT/gpurun_out/prof_cnn
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o cnn -- python3 $ROOT/scripts/cnn_timing.py 96 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $OUT/pmc -o cnn -- python3 $ROOT/scripts/cnn_timing.py 96 > $OUT/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d $OUT/pmc2 -o cnn -- python3 $ROOT/scripts/cnn_timing.py 96 > $OUT/pmc2.log 2>&1
cd $ROOT
tail -6 $OUT/stats.log
python3 scripts/pmc_kernels.py $OUT/pmc > $OUT/pmc_kernels.txt 2>&1
python3 scripts/pmc_kernels.py $OUT/pmc2 > $OUT/pmc2_kernels.txt 2>&1
grep -h "fs_k_vn\|igemm\|TOTAL\|MFMA" $OUT/pmc_kernels.txt $OUT/pmc2_kernels.txt
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -r head -8
mkdir -p $ROOT/gpurun_out/cnn_summary
python3 - $OUT/stats $ROOT/gpurun_out/cnn_summary/${TAG}_cnn_kernel_stats.csv <<'PY'
import csv, os, sqlite3, sys
db = [os.path.join(r, f) for r, _, fs in os.walk(sys.argv[1]) for f in fs if f.endswith(".db")][0]
con = sqlite3.connect(db)
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel (scripts/cnn_timing.py 96: 25 forwards each of module graph / folded MIOpen / hand-written)", "calls", "total_us", "average_us", "percent_of_gpu_time"])
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if pct >= 0.2:
            w.writerow([name[:120], calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.2f}"])
PY
cat $OUT/pmc_kernels.txt $OUT/pmc2_kernels.txt > $ROOT/gpurun_out/cnn_summary/${TAG}_cnn_pmc.txt
grep SpatialValueNet $OUT/stats.log > $ROOT/gpurun_out/cnn_summary/${TAG}_cnn_timing.txt
find $OUT -name "*.db" -delete
