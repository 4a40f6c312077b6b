"""Development helper: cloths far beyond the reference's sizes (150x150 = 22 500 particles; 300x250 = 75 000 particles, more
than the 16-bit rest-neighbour ids can address, so the search falls back to rest-position tests) -- streaming back-end vs the
CPU oracle, bit for bit, a few steps of a released sheet with a crumpled corner."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import cloth_params
from flingbot_amd import sim as fsim
from oracle import OracleSim
for dx, dz, steps in ((150, 150, 4), (300, 250, 2)):
    ctx = fsim.FlingSim(n_envs=1, solver=0)
    hip, orc = ctx.env(0), OracleSim()
    t0 = time.time()
    for s in (hip, orc):
        s.set_scene(cloth_params(dx, dz, pos=(0.0, -0.2, 0.0)))
        r = np.random.RandomState(7)
        p = s.get_positions().reshape(-1, 4).copy()
        k = min(3000, p.shape[0])
        p[:k, :3] = (r.rand(k, 3) * [0.12, 0.08, 0.12] + [0, 0.03, 0]).astype(np.float32)   # a heap: many contacts
        s.set_positions(p.ravel())
        s.step(steps)
    ok = np.array_equal(hip.get_positions().view(np.uint32), orc.get_positions().view(np.uint32))
    ok = ok and np.array_equal(hip.get_velocities().view(np.uint32), orc.get_velocities().view(np.uint32))
    ch, lh = ctx.get_last_neighbors(0); co, lo = orc.get_last_neighbors()
    mask = np.arange(96)[None, :] < co[:, None]
    ok = ok and np.array_equal(ch, co) and np.array_equal(np.where(mask, lh, -1), np.where(mask, lo, -1))
    print("%dx%d (%d particles), %d steps: %s, contacts max %d mean %.2f, %.0f s" % (
        dx, dz, dx * dz, steps, "ok" if ok else "MISMATCH", co.max(), co.mean(), time.time() - t0), flush=True)
    ctx.close()
