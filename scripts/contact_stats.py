"""Development helper: distribution of particle-contact candidate counts in the bench scenario."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from flingbot_amd import sim as fsim
E = 8
ctx = fsim.FlingSim(n_envs=E, solver=2)
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
for k in range(14):
    ctx.step(10)
    out = []
    for e in (0, 3, 7):
        cnt, _ = ctx.get_last_neighbors(e)
        out.append("%4d with contacts, mean %.2f max %d" % ((cnt > 0).sum(), cnt.mean(), cnt.max()))
    print("step %3d: " % (10 * (k + 1)) + " | ".join(out))
