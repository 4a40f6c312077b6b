"""Development helper: per-episode cost of the fused kernel.  A launch of the bench's 256 episodes lasts as long as its slowest
workgroup, so each listed seed is run alone (one workgroup) on the shipped library: ms per frame over frames 81..100."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from flingbot_amd import sim as fsim

seeds = [int(a) for a in sys.argv[1:]] or [0, 148, 103, 199, 213, 143, 232, 185]
for seed in seeds:
    ctx = fsim.FlingSim(n_envs=1, solver=2)
    bench.setup_episode(ctx.env(0), seed)
    ctx.step(80); ctx.sync()
    t0 = time.perf_counter()
    ctx.step(20); ctx.sync()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    cnt, _ = ctx.get_last_neighbors(0)
    print("seed %3d: %.3f ms per frame alone; contacts mean %.2f max %d, particles with any %d" % (seed, ms, cnt.mean(), cnt.max(), (cnt > 0).sum()))
    ctx.close()
