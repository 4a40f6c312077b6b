"""Development helper: wall time of the batched fling primitive (64x64 cloth, E episodes) on the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from fling_helpers import picker_centres
from flingbot_amd import sim as fsim
from flingbot_amd.primitives import FlingPrimitives

E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = fsim.FlingSim(n_envs=E, solver=0)
xs = (np.arange(64) - 31.5) * 0.00625
xx, zz = np.meshgrid(xs, xs)
for e in range(E):
    env = ctx.env(e)
    env.set_scene(bench.scene_params())
    env.step(1)
    w = env.get_positions().reshape(-1, 4)[0, 3]
    pos = np.zeros((4096, 4), np.float32)
    pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 3] = xx.ravel(), 0.0125, zz.ravel(), w
    env.set_positions(pos.ravel()); env.set_velocities(np.zeros(3 * 4096, np.float32))
    for c in picker_centres():
        env.add_sphere(0.02, c, [1, 0, 0, 0])
    st = np.array(env.get_shape_states()).reshape(-1, 14)
    for i, c in enumerate(picker_centres()):
        st[i] = np.hstack([c, c, [1, 0, 0, 0], [1, 0, 0, 0]])
    env.set_shape_states(st)
    ctx.picker_reset(e)
rng = np.random.default_rng(0)
p1 = np.stack([xs[rng.integers(0, 6, E)], np.zeros(E), xs[rng.integers(0, 6, E)]], 1)
p2 = np.stack([xs[63 - rng.integers(0, 6, E)], np.zeros(E), xs[rng.integers(0, 6, E)]], 1)
prim = FlingPrimitives(ctx, range(E))
ctx.sync(); t0 = time.perf_counter()
out = prim.pick_and_fling(p1, p2, [True] * E, [True] * E)
stable, steps = ctx.wait_until_stable(range(E), max_steps=300)
ctx.sync(); dt = time.perf_counter() - t0
total = prim.sim_steps + int(steps.sum())
print("E=%d: pick_and_fling + wait_until_stable %.2f s, %d simulation steps in total -> %.0f steps/s, %.2f flings/s; "
      "coverage %.3f; terminated %d" % (E, dt, total, total / dt, E / dt, float(np.mean(ctx.coverage())),
                                        sum(o["terminated"] for o in out)))
