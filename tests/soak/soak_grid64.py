"""Development helper: randomized parity soak of the grid forms -- fs_k_fused_grid64 (64-wide cloths, FUSED) and
fs_k_iterate_gridl (any canonical grid cloth, STREAM) -- against the CPU oracle, bit for bit (positions, velocities,
neighbour lists): crumples, loose heaps (long neighbour lists), tilted sheets hitting the ground, two-picker flings
(inverse mass 0 next to the fast spring form), cloths with fewer than 64 rows, per-type stiffnesses, several episodes per
launch.  Not part of the test-suite (minutes of oracle time).    python tests/soak/soak_grid64.py [cases]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenarios as sc
from conftest import cloth_params
from flingbot_amd import sim as fsim
from oracle import OracleSim

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(2024)
bad = 0
t0 = time.time()


def setup(s, kind, dimx, dimz, seed):
    r = np.random.RandomState(seed)
    stiff = (0.6 + 0.4 * (seed % 7) / 7, 0.7 + 0.3 * (seed % 5) / 5, 0.8)
    if kind == 0:
        sc.scenario_crumple(s, dimx, dimz, seed=seed, lift_steps=12, settle_steps=18)
    elif kind == 1:
        s.set_scene(cloth_params(dimx, dimz, pos=(0.0, -0.05, 0.0), stiff=stiff, mass=0.2 + (seed % 9) * 0.2))
        p = s.get_positions().reshape(-1, 4).copy()
        p[:, :3] = (r.rand(p.shape[0], 3) * [0.15, 0.1, 0.15] + [0, 0.05, 0]).astype(np.float32)  # loose heap
        s.set_positions(p.ravel()); s.set_velocities(np.zeros(3 * p.shape[0], np.float32))
        s.step(12)
    elif kind == 2:
        s.set_scene(cloth_params(dimx, dimz, pos=(0.05, -0.3, -0.1), stiff=stiff))
        p = s.get_positions().reshape(-1, 4).copy()
        ang = r.rand() * 3.1
        x, z = p[:, 0].copy() - 0.05, p[:, 2].copy() + 0.1
        p[:, 0] = x * np.cos(ang); p[:, 1] = 0.3 + x * np.sin(ang) + 0.5 * z  # tilted sheet falling on the ground
        s.set_positions(p.ravel()); s.set_velocities(np.zeros(3 * p.shape[0], np.float32))
        s.step(40)
    else:
        sc.scenario_fling(s, dimx, dimz, lift=0.2 + 0.1 * r.rand(), fling_dist=0.1 + 0.1 * r.rand(), settle_steps=8)


for case in range(n_cases):
    dimx, dimz = [(64, 64), (64, 40), (64, 17), (80, 72), (64, 5), (64, 64), (104, 64), (64, 33)][case % 8]
    solver = [fsim.FS_SOLVER_FUSED, fsim.FS_SOLVER_STREAM][case % 2] if dimx == 64 else fsim.FS_SOLVER_STREAM
    kind = (case // 2) % 4
    if kind == 3 and dimz < 8:
        kind = 2
    n_envs = 1 if kind in (0, 3) else 1 + case % 3
    seeds = [int(rng.randint(1 << 30)) for _ in range(n_envs)]
    ctx = fsim.FlingSim(n_envs=n_envs, solver=solver)
    orcs = [OracleSim() for _ in range(n_envs)]
    th = [threading.Thread(target=setup, args=(orcs[e], kind, dimx, dimz, seeds[e])) for e in range(n_envs)]
    [t.start() for t in th]
    if n_envs == 1:
        setup(ctx.env(0), kind, dimx, dimz, seeds[0])
    else:  # batched: replicate the scripted set-up calls per episode, stepping all episodes together
        class Batch:  # the scenarios used for n_envs > 1 only call set_scene / get/set_positions / set_velocities / step
            pass
        steps = {1: 12, 2: 40}[kind]
        for e in range(n_envs):
            class One:
                def __init__(self, env): self.env = env
                def __getattr__(self, name): return getattr(self.env, name)
                def step(self, n=1): pass
            setup(One(ctx.env(e)), kind, dimx, dimz, seeds[e])
        ctx.step(steps)
    [t.join() for t in th]
    form = ctx.last_kernel_form()
    ok = True
    cmax = 0
    for e in range(n_envs):
        okp = np.array_equal(ctx.get_positions(e).view(np.uint32), orcs[e].get_positions().view(np.uint32)) and \
            np.array_equal(ctx.get_velocities(e).view(np.uint32), orcs[e].get_velocities().view(np.uint32))
        ch, lh = ctx.get_last_neighbors(e); co, lo = orcs[e].get_last_neighbors()
        mask = np.arange(96)[None, :] < co[:, None]
        okn = np.array_equal(ch, co) and np.array_equal(np.where(mask, lh, -1), np.where(mask, lo, -1))
        ok = ok and okp and okn
        cmax = max(cmax, int(co.max()))
    bad += not ok
    print("case %2d %dx%d x%d solver %d form %d kind %d: %s (contacts max %d)" % (case, dimx, dimz, n_envs, solver, form, kind,
          "ok" if ok else "MISMATCH", cmax), flush=True)
    ctx.close()
print("soak grid forms: %d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
