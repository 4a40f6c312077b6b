"""Development helper: what a phase of the fused bench kernel costs, by ablation on a FIXED state.
    python scripts/ablation_timing.py save /tmp/state.npz            # the shipped library: 256 episodes to frame 80, state saved
    FLINGSIM_LIB=variants/libfs_<x>.so python scripts/ablation_timing.py time /tmp/state.npz
`time` loads the state and times 2 frames five times, the state put back in between: a library with one phase compiled
out (a local edit -- such switches give wrong results by design and are not kept in the tree) then shows that phase's share
of the frame on the very same cloths; any other variant library can be compared the same way (EXPERIMENTS.md R3.11)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from flingbot_amd import sim as fsim

mode, path = sys.argv[1], sys.argv[2]
E = 256
ctx = fsim.FlingSim(n_envs=E, solver=2)
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
if mode == "save":
    for _ in range(80):
        ctx.step(1)
    ctx.sync()
    np.savez(path, pos=np.stack([ctx.get_positions(e) for e in range(E)]), vel=np.stack([ctx.get_velocities(e) for e in range(E)]))
    print("saved", path)
else:
    g = np.load(path)
    out = []
    for rep in range(5):
        for e in range(E):
            ctx.set_positions(e, g["pos"][e]); ctx.set_velocities(e, g["vel"][e])
        ctx.sync(); ctx.timer_start(); ctx.step(1); ctx.step(1); out.append(ctx.timer_stop() / 2)
    print("%-40s %.3f ms per launch (min of 5: %.3f)" % (os.environ.get("FLINGSIM_LIB", "shipped"), float(np.median(out)), min(out)))
