#!/bin/bash
# EXPERIMENTS R4.2, second pass: the tiled boundary with XCD-local coherence only (variants/libfs_bwlight.so) against fs_k_boundary
mkdir -p gpurun_out/bw
FLINGSIM_LIB=variants/libfs_bwlight.so python -m pytest tests/test_shipped_kernels_gpu.py -q -k "boundary_forms or large_cloth_104" 2>&1 | tail -3 | tee gpurun_out/bw/tests2.txt
for r in 1 2; do
  FLINGSIM_BOUNDARY_WIDE=0 python scripts/eval_wall_breakdown.py 384 192 3 1 2>&1 | grep -E "tasks /" | sed "s/^/one-wg run $r: /" | tee -a gpurun_out/bw/eval384_2.txt
  FLINGSIM_LIB=variants/libfs_bwlight.so FLINGSIM_BOUNDARY_WIDE=1 python scripts/eval_wall_breakdown.py 384 192 3 1 2>&1 | grep -E "tasks /" | sed "s/^/light-wide run $r: /" | tee -a gpurun_out/bw/eval384_2.txt
done
for c in "104 16" "104 64" "104 128" "80 64"; do
  FLINGSIM_BOUNDARY_WIDE=0 python scripts/large_cloth_timing.py $c 2>&1 | grep cloth | sed "s/^/one-wg: /" | tee -a gpurun_out/bw/large2.txt
  FLINGSIM_LIB=variants/libfs_bwlight.so FLINGSIM_BOUNDARY_WIDE=1 python scripts/large_cloth_timing.py $c 2>&1 | grep cloth | sed "s/^/light-wide: /" | tee -a gpurun_out/bw/large2.txt
done
