#!/bin/bash
# Developer helper: build the current csrc tree into variants/libfs_<name>.so (extra flags after the name).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
mkdir -p "$ROOT/variants"
cd "$ROOT/flingbot_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 "$@" -shared \
    -o "$ROOT/variants/libfs_$NAME.so" fs_capi.hip fs_solver.hip fs_render.hip fs_picker.hip fs_loops.hip fs_image.hip fs_action.hip fs_valuenet.hip fs_observe.hip fs_hostapi.hip fs_scene.cpp
