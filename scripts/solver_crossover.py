"""Development helper: fused vs streaming back-end on the bench scenario (crumpled 64x64 cloth) for several batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from flingbot_amd import sim as fsim

SIZES = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 128, 256, 512]
for E in SIZES:
    row = []
    for solver in (2, 5, 1):   # grid-64 fused, dictionary-coded fused, streaming
        ctx = fsim.FlingSim(n_envs=E, solver=solver)
        for e in range(E):
            bench.setup_episode(ctx.env(e), e)
        ctx.step(60); ctx.sync()
        ctx.timer_start(); ctx.step(10); ms = ctx.timer_stop() / 10
        row.append(ms)
        ctx.close()
    print("E=%3d: fused grid-64 %.2f ms/step (%.0f steps/s)   fused coded %.2f (%.0f)   streaming %.2f ms/step (%.0f steps/s)" % (
        E, row[0], E / row[0] * 1e3, row[1], E / row[1] * 1e3, row[2], E / row[2] * 1e3), flush=True)
