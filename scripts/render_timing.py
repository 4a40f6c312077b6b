"""Development helper: fs_render / fs_observe wall time per call (720x720, one crumpled 64x64 cloth episode + the two pickers);
under rocprofv3 this is the launch list of BASELINE.json configs[1] (scripts/profile_render.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from flingbot_amd import sim as fsim
ctx = fsim.FlingSim(n_envs=1, solver=0)
bench.setup_episode(ctx.env(0), 0)
for c in ((0.5, 0.5, -0.5), (-0.5, 0.5, -0.5)):
    ctx.env(0).add_sphere(0.02, c, [1, 0, 0, 0])
ctx.step(80)
ctx.render(0); ctx.observe(0, 400); ctx.sync()
t0 = time.perf_counter()
for _ in range(20):
    rgba, depth = ctx.render(0)
dt = (time.perf_counter() - t0) / 20
print("fs_render 720x720: %.3f ms per call (includes the 4 MB device-to-host copies)" % (dt * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    ctx.observe(0, 400)
ctx.sync()
dt = (time.perf_counter() - t0) / 20
print("render + fs_observe 720 -> 400 on the device: %.3f ms per call" % (dt * 1e3))
