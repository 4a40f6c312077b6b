"""Development helper: one movep call (lift by 0.25 m at 5e-3 per step = 50 simulation steps) driven (a) from Python
through the pyflex-shaped accessors like the reference does, (b) by fs_movep on the device; 1 and 64 episodes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from flingbot_amd import sim as fsim
from oracle.picker import OraclePicker   # the restated host loop, used here only as the "Python caller" stand-in
import scenarios as sc
from conftest import cloth_params

def setup(ctx, e):
    env = ctx.env(e); env.set_scene(cloth_params(64, 64, pos=(0, 2.0, 0))); env.step(1)
    w = env.get_positions().reshape(-1, 4)[0, 3]
    p = sc.flat_positions(64, 64, y=0.0125, inv_mass=w); env.set_positions(p.ravel()); env.set_velocities(np.zeros(3 * 4096, np.float32))
    c0, c1 = p[0, :3].astype(np.float64) + [0, 0.02, 0], p[63, :3].astype(np.float64) + [0, 0.02, 0]
    return env, c0, c1

for E in (1, 64):
    ctx = fsim.FlingSim(n_envs=E)
    tools, tg = [], []
    for e in range(E):
        env, c0, c1 = setup(ctx, e)
        t = OraclePicker(env); t.reset([c0, c1]); tools.append(t); tg.append([c0 + [0, 0.25, 0], c1 + [0, 0.25, 0]])
        ctx.picker_reset(e)
    ctx.sync(); t0 = time.perf_counter()
    for e in range(E): tools[e].movep(tg[e], [True, True], speed=5e-3)
    ctx.sync(); dt_py = time.perf_counter() - t0
    back = [[np.array(a) - [0, 0.25, 0] for a in pair] for pair in tg]
    ctx.sync(); t0 = time.perf_counter()
    it = ctx.movep(range(E), back, [[1, 1]] * E, speed=5e-3)
    ctx.sync(); dt_dev = time.perf_counter() - t0
    print(f"E={E}: python-driven movep {dt_py*1e3:.1f} ms ({50*E/dt_py:.0f} steps/s) | fs_movep {dt_dev*1e3:.1f} ms ({int(it.sum())/dt_dev:.0f} steps/s)", flush=True)
