"""Development helper: launches dominated by the per-substep fixed work (predict, hash, neighbour search, contact set):
the bench scenario in its crumpled state with ONE solver iteration per substep.  For rocprofv3 --pmc runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from flingbot_amd import sim as fsim

E = 256
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = fsim.FlingSim(n_envs=E, solver=2)
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
ctx.step(80)
for e in range(E):
    tab = ctx.get_params(e); tab[0] = iters; ctx.set_params(e, tab)
ctx.sync(); ctx.timer_start(); ctx.step(20); ms = ctx.timer_stop() / 20
print("iterations=%d: %.3f ms/launch" % (iters, ms))
