#!/bin/bash
# Developer build with per-section shader-clock instrumentation (FS_TIMING) -> variants/libfs_timing.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/variants"
cd "$ROOT/flingbot_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -DFS_TIMING $FS_EXTRA -shared \
    -o "$ROOT/variants/libfs_timing.so" fs_capi.hip fs_solver.hip fs_render.hip fs_picker.hip fs_loops.hip fs_image.hip fs_action.hip fs_valuenet.hip fs_observe.hip fs_hostapi.hip fs_scene.cpp
