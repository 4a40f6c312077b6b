#!/bin/bash
# EXPERIMENTS R4.1: v_rsq_f32 in the spring / contact loops (variants/libfs_hwrsq.so, built with -DFS_HW_RSQ) against the
# shipped library, alternating on one box: headline, 64-episode entry, eval_loop.continuous -> gpurun_out/rsq/
mkdir -p gpurun_out/rsq
python scripts/rsq_dump.py shipped
FLINGSIM_LIB=variants/libfs_hwrsq.so python scripts/rsq_dump.py hwrsq
python scripts/rsq_dump.py compare shipped hwrsq | tee gpurun_out/rsq/compare.txt
for round in 1 2; do
  python bench.py --steps 50 --no-cpu-baseline --no-parity > gpurun_out/rsq/shipped_$round.json 2> gpurun_out/rsq/shipped_$round.err
  FLINGSIM_LIB=variants/libfs_hwrsq.so python bench.py --steps 50 --no-cpu-baseline --no-parity > gpurun_out/rsq/hwrsq_$round.json 2> gpurun_out/rsq/hwrsq_$round.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/rsq/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "headline %.0f (%.3f ms)" % (j["value"], j["roofline"]["kernel_ms_per_launch"]), "e64 %.0f" % j["configs"][1]["value"],
              "eval32 %.2f" % j["eval_loop"]["flings_per_s"], "cont %.2f" % j["eval_loop"]["continuous"]["flings_per_s"])
    except Exception as e:
        print(f, "ERR", e)
PY
