"""Development helper: fused vs streaming back-end on the bench scenario (crumpled 64x64 cloth) for several batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from flingbot_amd import sim as fsim

for E in (1, 8, 32, 64, 128, 256, 512):
    row = []
    for solver in (2, 1):
        ctx = fsim.FlingSim(n_envs=E, solver=solver)
        for e in range(E):
            bench.setup_episode(ctx.env(e), e % 16)
        ctx.step(60); ctx.sync()
        ctx.timer_start(); ctx.step(10); ms = ctx.timer_stop() / 10
        row.append(ms)
        ctx.close()
    print("E=%3d: fused %.2f ms/step (%.0f steps/s)   streaming %.2f ms/step (%.0f steps/s)" % (
        E, row[0], E / row[0] * 1e3, row[1], E / row[1] * 1e3), flush=True)
