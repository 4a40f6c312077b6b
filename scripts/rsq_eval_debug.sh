#!/bin/bash
mkdir -p gpurun_out/rsq
FLINGSIM_LIB=variants/libfs_hwrsq.so python - > gpurun_out/rsq/eval_debug.txt 2>&1 <<'PY'
import json, bench
print(json.dumps(bench.eval_loop_leg(0, episodes=32, actions=3, stream_tasks=384), indent=1))
PY
tail -30 gpurun_out/rsq/eval_debug.txt
