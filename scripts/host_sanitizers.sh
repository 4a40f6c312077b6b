#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the library's HOST code (GPU sanitizers are not available on this pool):
# fs_hostapi.hip (host-only entry points), fs_scene.cpp (scene / topology / adjacency builder), fs_tenants.cpp (co-tenant table)
# compiled with g++ as plain C++ into a host-only library, then the CPU tests that go through those entry points run against it
# (FLINGSIM_LIB) with the sanitizer runtimes preloaded, and the tenant table is exercised with a child process that gets killed.
# Runs here, no GPU.  Round 6 found one thing this way: memcpy(dst, NULL, 0) from an empty vector (the springs of a 1 x 1 cloth).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=${TMPDIR:-/tmp}/flingsim_asan; mkdir -p $OUT
g++ -g -O1 -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -w \
    -x c++ $ROOT/flingbot_amd/csrc/fs_hostapi.hip $ROOT/flingbot_amd/csrc/fs_scene.cpp $ROOT/flingbot_amd/csrc/fs_tenants.cpp \
    $ROOT/scripts/sanitize/err_shim.cpp -o $OUT/libfs_host_asan.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
cd $ROOT
FLINGSIM_LIB=$OUT/libfs_host_asan.so python -m pytest tests/test_oracle_cpu.py -q -W ignore -s \
    -k "host_scene or camera_matrices or prebuilt or sphere_mesh_pinned or derived_tables or envutils or fling_primitive_host or task_generator or action_selector_host" \
    > $OUT/run.log 2>&1 || { tail -40 $OUT/run.log; exit 1; }
tail -1 $OUT/run.log
python scripts/sanitize/tenants_under_sanitizers.py $OUT/libfs_host_asan.so
# the checker too: the C oracle (solver step + rasteriser) built with the same sanitizers, on its physics / collideShapes / picker
# tests and on the crumple + scripted-fling scenarios of the fixture kit
env -u LD_PRELOAD gcc -g -O1 -fPIC -std=c11 -ffp-contract=off -fno-fast-math -mfma -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o $OUT/liboracle_asan.so $ROOT/oracle/flex_oracle.c $ROOT/oracle/raster_oracle.c -lm
FLINGSIM_ORACLE_LIB=$OUT/liboracle_asan.so python -m pytest tests/test_oracle_cpu.py tests/test_external_fixtures.py -q -W ignore -s -m "not gpu" \
    -k "grid_counts or grid_3x2 or parameter_table or free_fall or pinned_particle or ground_contact or spring_pair or self_collision or sphere_pushes or deterministic or collide_shapes or observe_oracle or picker_restatement or coverage_oracle or ingest_path" \
    >> $OUT/run.log 2>&1 || { tail -40 $OUT/run.log; exit 1; }
tail -1 $OUT/run.log
if grep -q "runtime error\|AddressSanitizer" $OUT/run.log; then grep -A6 "runtime error\|AddressSanitizer" $OUT/run.log | head -40; exit 1; fi
echo "host code clean under ASan + UBSan"
