"""Development helper: two forms of the streaming back-end side by side on the bench scenario (crumpled 64x64 cloths, frames
80..110) and on 104x104 heaps.  usage: boundary_timing.py [solver_a solver_b]   (default 7 1: separate boundary kernels
against fs_k_boundary; 9 1: grid-L against grid-T iterate)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from conftest import cloth_params
from flingbot_amd import sim as fsim

SA, SB = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7, 1)

def time_ctx(ctx, warm, steps):
    ctx.step(warm); ctx.sync()
    ctx.timer_start(); ctx.step(steps); return ctx.timer_stop() / steps

for E in (1, 8, 32, 64, 128, 256):
    row = []
    for solver in (SA, SB):
        ctx = fsim.FlingSim(n_envs=E, solver=solver)
        for e in range(E):
            bench.setup_episode(ctx.env(e), e % 16)
        row.append(time_ctx(ctx, 80, 30))
        ctx.close()
    print("64x64 x %3d: solver %d %.3f ms/step (%.0f steps/s)   solver %d %.3f ms/step (%.0f steps/s)" % (
        E, SA, row[0], E / row[0] * 1e3, SB, row[1], E / row[1] * 1e3), flush=True)
for E in (1, 16, 64):
    row = []
    for solver in (SA, SB):
        ctx = fsim.FlingSim(n_envs=E, solver=solver)
        p = cloth_params(104, 104, pos=(0.0, -0.15, 0.0))
        for e in range(E):
            ctx.set_scene(e, p)
            rng = np.random.RandomState(100 + e % 8)
            pos = ctx.get_positions(e).reshape(-1, 4).copy()
            pos[:, :3] += (rng.randn(pos.shape[0], 3) * 0.003).astype(np.float32)
            pos[:2500, :3] = (rng.rand(2500, 3) * [0.12, 0.05, 0.12] + [0.0, 0.02, 0.0]).astype(np.float32)
            ctx.set_positions(e, pos.ravel())
        row.append(time_ctx(ctx, 10, 10))
        ctx.close()
    print("104x104 x %3d: solver %d %.3f ms/step   solver %d %.3f ms/step" % (E, SA, row[0], SB, row[1]), flush=True)
