"""Development helper: step time of cloths that do not fit the fused LDS kernel (N > 4096): streaming back-end."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import cloth_params
from flingbot_amd import sim as fsim

cases = ((64, 64), (80, 64), (104, 64), (104, 16), (104, 1))
if len(sys.argv) > 2:
    cases = ((int(sys.argv[1]), int(sys.argv[2])),)
ell = os.environ.get("FS_STREAM_ELL") == "1"  # uncompressed adjacency, for comparison
for dim, E in cases:
    ctx = fsim.FlingSim(n_envs=E, solver=4 if ell else (1 if dim == 64 else 0))
    for e in range(E):
        ctx.set_scene(e, cloth_params(dim, dim, pos=(0.0, -0.3, 0.0)))
    ctx.step(3); ctx.sync()
    ctx.timer_start(); ctx.step(10); ms = ctx.timer_stop() / 10
    print("%3dx%-3d cloth, %3d episodes, %s: %.2f ms/step -> %.0f episode-steps/s" % (
        dim, dim, E, "streaming, ELL adjacency" if ell else "streaming", ms, E / ms * 1e3), flush=True)
    ctx.close()
