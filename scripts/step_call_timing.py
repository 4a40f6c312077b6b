import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
from flingbot_amd import sim as fsim
E = 64
ctx = fsim.FlingSim(n_envs=E, solver=0)
for e in range(E): bench.setup_episode(ctx.env(e), e)
ctx.step(80); ctx.sync()
for mode in ("step(1) x 40", "step(40)"):
    ctx.sync(); t0 = time.perf_counter(); ctx.timer_start()
    if mode == "step(40)": ctx.step(40)
    else:
        for _ in range(40): ctx.step(1)
    ms = ctx.timer_stop(); ctx.sync(); wall = time.perf_counter() - t0
    print("%s: gpu %.3f ms/step, wall %.3f ms/step, groups %d" % (mode, ms / 40, wall / 40 * 1e3, ctx.last_stream_groups()), flush=True)
