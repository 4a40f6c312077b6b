#!/bin/bash
# Profiles the default bench workload with rocprofv3 (kernel trace + stats, then separate PMC passes as the MI355X guide
# prescribes).  Run on the GPU box from the repo root:  bash scripts/profile_bench.sh [tag]
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-parity"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o sq -- python3 $ARGS > $OUT/bench_sq.log 2>&1
# memory instructions (scratch = spill traffic shows up in the VMEM / FLAT counts; EXPERIMENTS.md R3.4)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAVES -d $OUT/pmc_mem -o mem -- python3 $ARGS > $OUT/bench_mem.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -30
