"""Development helper: shader-clock totals per kernel section (FS_TIMING build, variants/libfs_timing.so) for one launch
of the bench scenario in its crumpled state.  Run with FLINGSIM_LIB=variants/libfs_timing.so."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from flingbot_amd import sim as fsim

E = 256
ctx = fsim.FlingSim(n_envs=E, solver=int(os.environ.get("FS_SOLVER", "2")))
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
os.environ["FS_QUIET"] = "1"
fd = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY)
os.dup2(devnull, 1)          # silence the warm-up launches' device printf
ctx.step(80); ctx.sync()
sys.stdout.flush(); os.dup2(fd, 1)
ctx.step(1); ctx.sync()
