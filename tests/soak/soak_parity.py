"""Development helper: randomized parity soak -- many seeds / cloth sizes / both back-ends against the CPU oracle,
bit for bit (positions, velocities, neighbour lists).  Not part of the test-suite (minutes of oracle time)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenarios as sc
from conftest import cloth_params
from flingbot_amd import sim as fsim
from oracle import OracleSim

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(123)
bad = 0
t0 = time.time()
for case in range(n_cases):
    dimx, dimz = [(24, 24), (32, 48), (64, 64), (40, 17), (64, 40)][case % 5]
    solver = (1, 2, 9)[case % 3]   # streaming, fused, co-tenant mode (fused when the launch fits, round 6)
    seed = int(rng.randint(1 << 30))
    ctx = fsim.FlingSim(n_envs=1, solver=solver)
    hip, orc = ctx.env(0), OracleSim()
    kind = case % 4
    for s in (hip, orc):
        if kind == 0:
            sc.scenario_crumple(s, dimx, dimz, seed=seed, lift_steps=12, settle_steps=18)
        elif kind == 1:
            s.set_scene(cloth_params(dimx, dimz, pos=(0.0, -0.05, 0.0)))
            r = np.random.RandomState(seed)
            p = s.get_positions().reshape(-1, 4).copy()
            p[:, :3] = (r.rand(p.shape[0], 3) * [0.15, 0.1, 0.15] + [0, 0.05, 0]).astype(np.float32)  # loose heap
            s.set_positions(p.ravel()); s.set_velocities(np.zeros(3 * p.shape[0], np.float32))
            s.step(15)
        elif kind == 3:
            # a sheet falling through 1..8 kinematic spheres, some of them moving (previous != current pose: the sweep), some
            # stacked so that the cap of 6 shape contacts per particle (collideShapes) bites
            s.set_scene(cloth_params(dimx, dimz, pos=(0.0, -0.14, 0.0)))
            r = np.random.RandomState(seed)
            k = 1 + seed % 8
            centres = r.rand(k, 3) * [0.2, 0.08, 0.2] + [0.0, 0.03, 0.0]
            if seed % 3 == 0:
                centres[:, :] = centres[0] + r.rand(k, 3) * 0.01      # all in one place: up to 8 candidates per particle
            for c in centres:
                s.add_sphere(0.02, c, [1, 0, 0, 0])
            st = np.array(s.get_shape_states(), np.float32).reshape(-1, 14)
            for _ in range(30):
                st[:, 3:6] = st[:, 0:3]
                st[:, 0:3] += (r.rand(k, 3).astype(np.float32) - 0.5) * 0.004
                s.set_shape_states(st.ravel())
                s.step(1)
        else:
            s.set_scene(cloth_params(dimx, dimz, pos=(0.0, -0.3, 0.0), stiff=(0.6 + 0.4 * (seed % 7) / 7, 0.9, 0.8)))
            r = np.random.RandomState(seed)
            p = s.get_positions().reshape(-1, 4).copy()
            ang = r.rand() * 3.1
            x, z = p[:, 0].copy(), p[:, 2].copy()
            p[:, 0] = x * np.cos(ang); p[:, 1] = 0.3 + x * np.sin(ang) + 0.5 * z  # tilted sheet falling on the ground
            s.set_positions(p.ravel()); s.set_velocities(np.zeros(3 * p.shape[0], np.float32))
            s.step(40)
    ok = np.array_equal(hip.get_positions().view(np.uint32), orc.get_positions().view(np.uint32)) and \
        np.array_equal(hip.get_velocities().view(np.uint32), orc.get_velocities().view(np.uint32))
    ch, lh = ctx.get_last_neighbors(0); co, lo = orc.get_last_neighbors()
    mask = np.arange(96)[None, :] < co[:, None]
    ok = ok and np.array_equal(ch, co) and np.array_equal(np.where(mask, lh, -1), np.where(mask, lo, -1))
    ok = ok and np.array_equal(ctx.get_last_shape_candidates(0), orc.get_last_shape_candidates())   # collideShapes, white box
    bad += not ok
    print("case %2d %dx%d solver %d kind %d seed %d: %s (contacts max %d)" % (case, dimx, dimz, solver, kind, seed,
          "ok" if ok else "MISMATCH", co.max()), flush=True)
    ctx.close()
print("soak: %d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
