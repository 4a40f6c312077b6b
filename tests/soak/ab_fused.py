"""A/B helper for fused-kernel variants: FLINGSIM_LIB=variants/libfs_<x>.so python tests/soak/ab_fused.py [episodes]
Prints ms/launch of the bench workload in its crumpled state (frames 80-180) and checks episodes against the oracle."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from flingbot_amd import sim as fsim
from oracle import OracleSim

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
solver = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tag = os.environ.get("FLINGSIM_LIB", "default")
# parity first (short): 4 episodes x 40 frames
ctx = fsim.FlingSim(n_envs=E, solver=solver)
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
sample = sorted({0, 1, E // 2, E - 1})
orcs = [OracleSim() for _ in sample]
def work(k):
    bench.setup_episode(orcs[k], sample[k]); orcs[k].step(40)
th = [threading.Thread(target=work, args=(k,)) for k in range(len(sample))]
[t.start() for t in th]
ctx.step(40)
[t.join() for t in th]
ok = all(np.array_equal(ctx.get_positions(s).view(np.uint32), o.get_positions().view(np.uint32)) and
         np.array_equal(ctx.get_velocities(s).view(np.uint32), o.get_velocities().view(np.uint32)) for s, o in zip(sample, orcs))
print(f"[{tag}] parity 40 frames, episodes {sample}: {'BIT-EXACT' if ok else 'MISMATCH'}", flush=True)
ctx.step(40)
ctx.sync()
res = []
for rep in range(3):
    ctx.timer_start(); ctx.step(30); res.append(ctx.timer_stop() / 30)
print(f"[{tag}] E={E} solver={solver} form={ctx.last_kernel_form()} ms/launch (frames 80-170, 3 x 30): " + " ".join("%.3f" % r for r in res) +
      f"  -> {E / min(res) * 1e3:.0f} steps/s", flush=True)
