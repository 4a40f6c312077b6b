"""Development helper: is a lone small cloth's step bound by the host's launch rate or by the device?  200 x fs_step of one
32 x 32 (and one 64 x 64) cloth: wall time until the launches are QUEUED (the calls return) vs until the device is done."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios as sc
from flingbot_amd import sim as fsim
for dim in (32, 64):
    ctx = fsim.FlingSim(n_envs=1)
    sc.canonical_flat(ctx.env(0), dim)
    ctx.step(20); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.step(1)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print("%dx%d: 200 steps queued in %.1f ms (%.3f ms per step on the host), device done %.1f ms later; total %.3f ms per step" % (
        dim, dim, (t1 - t0) * 1e3, (t1 - t0) * 5, (t2 - t1) * 1e3, (t2 - t0) * 5))
    ctx.close()
