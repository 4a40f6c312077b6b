#!/bin/bash
# EXPERIMENTS R4.1, second pass: hardware table dump + the clamped form's eval loop
mkdir -p gpurun_out/rsq
FLINGSIM_LIB=variants/libfs_hwrsq.so python scripts/rsq_table.py gpurun_out/rsq/v_rsq_f32_gfx950.npz 2>&1 | tee gpurun_out/rsq/table.txt
FLINGSIM_LIB=variants/libfs_hwrsq.so python bench.py --steps 50 --no-cpu-baseline --no-parity > gpurun_out/rsq/hwrsq_3.json 2> gpurun_out/rsq/hwrsq_3.err
python bench.py --steps 50 --no-cpu-baseline --no-parity > gpurun_out/rsq/shipped_3.json 2> gpurun_out/rsq/shipped_3.err
FLINGSIM_LIB=variants/libfs_hwrsq.so python bench.py --steps 50 --no-cpu-baseline --no-parity > gpurun_out/rsq/hwrsq_4.json 2> gpurun_out/rsq/hwrsq_4.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/rsq/*_[34].json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "headline %.0f (%.3f ms)" % (j["value"], j["roofline"]["kernel_ms_per_launch"]), "e64 %.0f" % j["configs"][1]["value"],
              j["eval_loop"].get("error") or ("eval32 %.2f cont %.2f" % (j["eval_loop"]["flings_per_s"], j["eval_loop"]["continuous"]["flings_per_s"])))
    except Exception as e:
        print(f, "ERR", e)
PY
