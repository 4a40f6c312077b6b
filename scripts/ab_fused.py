"""Development helper: A/B of library builds on one box -- the fused grid-64 kernel on bench.py's crumpled-sheet workload, with
and without two parked pickers.  usage: ab_fused.py lib1.so lib2.so ...   (each library runs in a child process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from flingbot_amd import sim as fsim
from conftest import cloth_params
E = int(os.environ.get("AB_E", "256"))
ctx = fsim.FlingSim(n_envs=E, solver=int(os.environ.get("AB_SOLVER", "2")))
for e in range(E):
    ctx.set_scene(e, cloth_params(64, 64, pos=(0.0, -0.3 - 0.001 * (e %% 7), 0.0)))
    p = ctx.get_positions(e).reshape(-1, 4).copy()     # vertical sheet, like bench.py's workload
    y, z = p[:, 1].copy(), p[:, 2].copy()
    p[:, 1] = 0.05 + (z - z.min()); p[:, 2] = 0.0 + 0.02 * np.sin(40 * p[:, 0])
    ctx.set_positions(e, p.ravel())
ctx.step(60); ctx.sync()
res = []
for rep in range(3):
    ctx.timer_start(); ctx.step(20); res.append(ctx.timer_stop() / 20)
print("  no pickers      : %%.4f ms per step (min of 3; all %%s)" %% (min(res), ["%%.4f" %% r for r in res]))
for e in range(E):
    for c in ((0.5, 0.5, -0.5), (-0.5, 0.5, -0.5)):
        ctx.add_sphere(e, 0.02, c, [1, 0, 0, 0])
ctx.step(2); ctx.sync()
res = []
for rep in range(3):
    ctx.timer_start(); ctx.step(20); res.append(ctx.timer_stop() / 20)
print("  2 parked pickers: %%.4f ms per step (min of 3; all %%s)" %% (min(res), ["%%.4f" %% r for r in res]))
''' % (ROOT, ROOT)
for rnd in range(2):
    for lib in sys.argv[1:]:
        print("==", lib, flush=True)
        env = dict(os.environ, FLINGSIM_LIB=os.path.abspath(lib))
        subprocess.run([sys.executable, "-c", CHILD], env=env)
