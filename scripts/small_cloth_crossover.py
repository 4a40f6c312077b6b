"""Development helper: fused (one workgroup per episode, one launch per frame) against streaming (129 launches per frame) for
SMALL cloths and small launches, where the streaming path sits on its launch-latency floor whatever the cloth size.
usage: small_cloth_crossover.py [dims...]   (crumpled sheets, ms per step)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenarios as sc
from flingbot_amd import sim as fsim

dims = [int(a) for a in sys.argv[1:]] or [16, 24, 32, 40, 48, 56, 64]
for dim in dims:
    for E in (1, 8, 32, 64):
        row = []
        for solver in (2, 1):
            ctx = fsim.FlingSim(n_envs=E, solver=solver)
            for e in range(E):
                env = ctx.env(e)
                env.set_scene(sc.cloth_params(dim, dim, pos=(0.0, -0.2, 0.0)))
                rng = np.random.RandomState(e)
                pos = env.get_positions().reshape(-1, 4).copy()
                pos[:, 1] = 0.02 + np.arange(dim * dim) // dim * 0.00625          # vertical sheet like bench.py's
                pos[:, 2] = 0.0
                pos[:, :3] += (rng.rand(dim * dim, 3).astype(np.float32) - 0.5) * 0.002
                env.set_positions(pos.ravel())
            ctx.step(60); ctx.sync()
            ctx.timer_start()
            for _ in range(20):
                ctx.step(1)
            row.append(ctx.timer_stop() / 20)
            ctx.close()
        print("%2dx%-2d (%4d particles) x %2d episodes: fused %.3f ms/step   streaming %.3f ms/step   -> %s" % (
            dim, dim, dim * dim, E, row[0], row[1], "fused" if row[0] < row[1] else "streaming"), flush=True)
