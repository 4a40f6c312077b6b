"""Development helper: per-section shader clocks of fs_k_fused_step for ONE small crumpled cloth (FS_TIMING build).
usage: FLINGSIM_LIB=variants/libfs_timing.so python scripts/small_section_timing.py [dim]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenarios as sc
from flingbot_amd import sim as fsim

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = fsim.FlingSim(n_envs=1, solver=2)
env = ctx.env(0)
env.set_scene(sc.cloth_params(dim, dim, pos=(0.0, -0.2, 0.0)))
pos = env.get_positions().reshape(-1, 4).copy()
pos[:, 1] = 0.02 + np.arange(dim * dim) // dim * 0.00625
pos[:, 2] = 0.0
pos[:, :3] += (np.random.RandomState(0).rand(dim * dim, 3).astype(np.float32) - 0.5) * 0.002
env.set_positions(pos.ravel())
fd = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY)
os.dup2(devnull, 1)
ctx.step(60); ctx.sync()
sys.stdout.flush(); os.dup2(fd, 1)
ctx.timer_start(); ctx.step(1); ms = ctx.timer_stop()
ctx.sync()
print("one frame: %.3f ms" % ms)
