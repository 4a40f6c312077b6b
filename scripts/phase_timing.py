"""Development helper: per-launch time of the fused kernel in different physical regimes (256 episodes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from flingbot_amd import sim as fsim
import scenarios as sc

E = 256
def timed(ctx, n):
    ctx.sync(); ctx.timer_start(); ctx.step(n); return ctx.timer_stop() / n

ctx = fsim.FlingSim(n_envs=E, solver=int(os.environ.get("FS_SOLVER", "2")))
for e in range(E):
    env = ctx.env(e); env.set_scene(bench.scene_params())
    w = env.get_positions().reshape(-1, 4)[0, 3]
    env.set_positions(sc.flat_positions(64, 64, y=1.5, inv_mass=w).ravel())
ctx.step(2)
print("A free fall (no contacts at all):      %.3f ms/launch" % timed(ctx, 20), flush=True)
for e in range(E):
    env = ctx.env(e); w = env.get_positions().reshape(-1, 4)[0, 3]
    env.set_positions(sc.flat_positions(64, 64, y=0.006, inv_mass=w).ravel()); env.set_velocities(np.zeros(3 * 4096, np.float32))
ctx.step(30)
print("B flat on the ground (plane contact):  %.3f ms/launch  maxv %.3g" % (timed(ctx, 20), np.abs(ctx.get_velocities(0)).max()), flush=True)
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
for k in range(12):
    ms = timed(ctx, 10)
    cnt, _ = ctx.get_last_neighbors(0)
    print("C bench scenario steps %3d-%3d: %.3f ms/launch  contacts mean %.2f max %d  maxv %.2f" % (
        10 * k, 10 * k + 10, ms, cnt.mean(), cnt.max(), np.abs(ctx.get_velocities(0)).max()), flush=True)
# split of one launch into per-substep fixed work (predict, hash, neighbour search, contact set, finalize) and
# per-iteration work: same crumpled state, iterations 30 vs 2
for iters in (30, 2):
    for e in range(E):
        tab = ctx.get_params(e); tab[0] = iters; ctx.set_params(e, tab)
    print("D crumpled, %2d iterations: %.3f ms/launch" % (iters, timed(ctx, 5)), flush=True)
