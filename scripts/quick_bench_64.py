"""Development helper: the 64-episodes-per-GPU figure of bench.py alone (streaming back-end, crumpled 64x64 cloths): steps/s over 100 frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from flingbot_amd import sim as fsim
ctx = fsim.FlingSim(n_envs=64, solver=0)
for e in range(64):
    bench.setup_episode(ctx.env(e), e)
for _ in range(70):
    ctx.step(1)
ctx.sync()
for rep in range(3):
    t0 = time.perf_counter(); ctx.timer_start()
    for _ in range(100):
        ctx.step(1)
    ms = ctx.timer_stop(); ctx.sync(); dt = time.perf_counter() - t0
    print("64 episodes: %.0f steps/s (wall), %.3f ms per step (GPU)" % (6400 / dt, ms / 100))
