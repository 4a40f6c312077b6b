"""Development helper: fs_observe against oracle/observe.py on many random scenes (cloth sizes, crumple seeds, gripper
spheres in view, render / observation sizes), bit for bit: observation tensor, largest-component mask, bounding box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import cloth_params
from flingbot_amd import sim as fsim
from oracle import observe as oo

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(11)
bad = 0
for case in range(n_cases):
    dx, dz = int(rng.randint(12, 70)), int(rng.randint(12, 70))
    render_dim = int(rng.choice([720, 480, 333, 256]))
    image_dim = int(rng.choice([400, 256, 128, 97, render_dim]))
    ctx = fsim.FlingSim(n_envs=1, solver=0)
    env = ctx.env(0)
    env.set_scene(cloth_params(dx, dz, pos=(float(rng.uniform(-0.2, 0.2)), -float(rng.uniform(0.05, 0.4)), float(rng.uniform(-0.2, 0.2)))))
    p = env.get_positions().reshape(-1, 4).copy()
    p[:, :3] += (rng.randn(p.shape[0], 3) * 0.005).astype(np.float32)
    env.set_positions(p.ravel())
    for _ in range(int(rng.randint(0, 3))):
        env.add_sphere(0.03, [float(rng.uniform(-0.6, 0.6)), 0.05, float(rng.uniform(-0.6, 0.6))], [1, 0, 0, 0])
    ctx.step(int(rng.randint(5, 60)))
    cp = ctx.get_camera_params(0)
    ctx.set_camera_params(0, [*cp[2:8], render_dim, render_dim])
    rgba, depth = ctx.render(0)
    obs, bbox, mask = ctx.observe(0, image_dim, want_mask=True)
    ref_obs, rgb, d, ref_mask, crop = oo.get_obs(rgba, depth, render_dim, image_dim)
    ok = np.array_equal(obs.cpu().numpy(), ref_obs)
    if ref_mask is None:
        ok = ok and bbox.tolist() == [-1, -1, -1, -1, 0]
    else:
        x, y = np.where(ref_mask)
        ok = ok and np.array_equal(mask.cpu().numpy(), ref_mask) and bbox.tolist() == [x.min(), x.max(), y.min(), y.max(), int(ref_mask.sum())]
    bad += not ok
    print("case %2d cloth %dx%d render %d -> %d: %s (component %d px, raw mask %d px)" % (
        case, dx, dz, render_dim, image_dim, "ok" if ok else "MISMATCH", int(bbox[4]), int(oo.cloth_mask_raw(rgb).sum())), flush=True)
    ctx.close()
print("soak_observe: %d cases, %d mismatches" % (n_cases, bad))
