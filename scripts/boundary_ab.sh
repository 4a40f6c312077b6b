#!/bin/bash
# EXPERIMENTS R4.8: fs_k_boundary with batched loads (this tree) against the previous library (variants/libfs_prev.so)
mkdir -p gpurun_out/bw
python -m pytest tests/test_shipped_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/bw/tests3.txt
for r in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export FLINGSIM_LIB=variants/libfs_prev.so; else unset FLINGSIM_LIB; fi
    python scripts/eval_wall_breakdown.py 384 192 3 1 2>&1 | grep -E "tasks /" | sed "s/^/$lib run $r: /" | tee -a gpurun_out/bw/eval384_3.txt
  done
done
for lib in prev new; do
  if [ $lib = prev ]; then export FLINGSIM_LIB=variants/libfs_prev.so; else unset FLINGSIM_LIB; fi
  python scripts/quick_bench_64.py 2>&1 | grep steps | sed "s/^/$lib: /" | tee -a gpurun_out/bw/e64_3.txt
  for c in "104 16" "104 64" "80 64"; do
    python scripts/large_cloth_timing.py $c 2>&1 | grep cloth | sed "s/^/$lib: /" | tee -a gpurun_out/bw/large3.txt
  done
done
