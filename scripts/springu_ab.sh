#!/bin/bash
# EXPERIMENTS R4.5: the spring scale as k (1 - L / |e|) = one fma on the reciprocal root, instead of len = l2 * inv, C = len - L,
# C * inv (variants/libfs_springu.so, -DFS_SPRING_U) against the shipped library, alternating on one box.
mkdir -p gpurun_out/springu
for r in 1 2; do
  python bench.py --steps 50 --no-cpu-baseline --no-parity --no-c2 --no-dropin > gpurun_out/springu/shipped_$r.json 2> /dev/null
  FLINGSIM_LIB=variants/libfs_springu.so python bench.py --steps 50 --no-cpu-baseline --no-parity --no-c2 --no-dropin > gpurun_out/springu/springu_$r.json 2> /dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/springu/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "headline %.0f (%.3f ms)" % (j["value"], j["roofline"]["kernel_ms_per_launch"]), "e64 %.0f" % j["configs"][1]["value"],
              j["eval_loop"].get("error") or ("eval32 %.2f cont %.2f" % (j["eval_loop"]["flings_per_s"], j["eval_loop"]["continuous"]["flings_per_s"])))
    except Exception as e:
        print(f, "ERR", e)
PY
