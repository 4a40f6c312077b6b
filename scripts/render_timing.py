"""Development helper: fs_render wall time per call (720x720, one 64x64 cloth episode, crumpled)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from flingbot_amd import sim as fsim
ctx = fsim.FlingSim(n_envs=1, solver=0)
bench.setup_episode(ctx.env(0), 0)
ctx.step(60)
ctx.render(0); ctx.sync()
t0 = time.perf_counter()
for _ in range(20):
    rgba, depth = ctx.render(0)
dt = (time.perf_counter() - t0) / 20
print("fs_render 720x720: %.2f ms per call (includes the 4 MB device-to-host copies)" % (dt * 1e3))
