#!/bin/bash
# EXPERIMENTS R4.2: fs_k_boundary_wide (tiles of 4096 particles) against fs_k_boundary (one workgroup per episode), same library,
# alternating on one box: the evaluation loop at 384 tasks / 192 slots and 64 / 32, and plain large-cloth launches.
mkdir -p gpurun_out/bw
python -m pytest tests/test_shipped_kernels_gpu.py -q -k "boundary_forms or large_cloth_104 or streaming_64" 2>&1 | tail -5 | tee gpurun_out/bw/tests.txt
for r in 1 2; do
  for w in 0 1; do
    FLINGSIM_BOUNDARY_WIDE=$w python scripts/eval_wall_breakdown.py 384 192 3 1 2>&1 | grep -E "tasks /|fs_advance calls|inside fs_advance" | sed "s/^/wide=$w run $r: /" | tee -a gpurun_out/bw/eval384.txt
  done
done
for w in 0 1; do
  FLINGSIM_BOUNDARY_WIDE=$w python scripts/eval_wall_breakdown.py 64 32 3 1 2>&1 | grep -E "tasks /" | sed "s/^/wide=$w: /" | tee -a gpurun_out/bw/eval64.txt
  for c in "104 16" "104 64" "104 128" "80 64" "80 128"; do
    FLINGSIM_BOUNDARY_WIDE=$w python scripts/large_cloth_timing.py $c 2>&1 | sed "s/^/wide=$w: /" | tee -a gpurun_out/bw/large.txt
  done
done
