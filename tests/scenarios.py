"""Scripted episodes used by the parity tests, the smoke test and bench.py.

Every scenario drives a duck-typed `sim` that exposes the reference's pyflex call names (set_scene, step,
get/set_positions, get/set_velocities, add_sphere, get/set_shape_states) -- either oracle.OracleSim or
flingbot_amd.sim.EnvView -- with identical fp32 inputs, so the two trajectories can be compared bit for bit.

The picker logic restates what the reference's callers do around the solver (environment/flex_utils.py:104-205
Picker._get_pos/_set_pos/step and environment/simEnv.py:739-769 movep): a kinematic sphere carries a pinned
(invMass = 0) particle by teleporting both by the same delta before each pyflex.step().
"""
import numpy as np

from conftest import cloth_params  # noqa: F401


def flat_positions(dimx, dimz, y=0.00625 * 2, inv_mass=None, spacing=0.00625):
    """Flat centred grid (mirrors flex_utils.py:398-415 set_to_flatten, but keeps the particle spacing)."""
    xs = (np.arange(dimx) - (dimx - 1) / 2.0) * spacing
    zs = (np.arange(dimz) - (dimz - 1) / 2.0) * spacing
    xx, zz = np.meshgrid(xs, zs)
    p = np.zeros((dimx * dimz, 4), np.float32)
    p[:, 0] = xx.ravel()
    p[:, 1] = y
    p[:, 2] = zz.ravel()
    p[:, 3] = inv_mass if inv_mass is not None else 1.0
    return p


class ScriptedPickers:
    """Two kinematic spheres; mirrors Picker/PickerPickPlace call order (flex_utils.py:74-252)."""

    def __init__(self, sim, radius=0.02, positions=((0.5, 0.5, -0.5), (-0.5, 0.5, -0.5))):
        self.sim = sim
        self.radius = radius
        for p in positions:
            sim.add_sphere(radius, p, [1, 0, 0, 0])
        st = np.array(sim.get_shape_states(), np.float32)
        sim.set_shape_states(st)
        self.picked = [None, None]
        self.saved_w = sim.get_positions().reshape(-1, 4)[:, 3].copy()

    def picker_pos(self):
        return np.array(self.sim.get_shape_states(), np.float32).reshape(-1, 14)[:, :3].copy()

    def grasp_nearest(self, k, threshold=0.005 + 0.02 + 0.00625):
        """flex_utils.py:147-165: nearest free particle within threshold of the picker centre."""
        pp = self.picker_pos()[k].astype(np.float64)
        pos = self.sim.get_positions().reshape(-1, 4).astype(np.float64)
        d = np.linalg.norm(pos[:, :3] - pp, axis=1)
        cand = np.where(d <= threshold)[0]
        cand = [c for c in cand if c not in self.picked]
        if cand:
            self.picked[k] = int(cand[int(np.argmin(d[cand]))])
        return self.picked[k]

    def release(self, k):
        self.picked[k] = None

    def move(self, deltas):
        """One Picker.step: shift spheres (prev := old current) and carried particles, pin them (w = 0)."""
        sim = self.sim
        deltas = np.asarray(deltas, np.float64).reshape(-1, 3)
        st = np.array(sim.get_shape_states(), np.float64).reshape(-1, 14)
        pos = sim.get_positions().reshape(-1, 4).astype(np.float64)
        new_pos = pos.copy()
        new_pos[:, 3] = self.saved_w  # un-picked particles get their mass back
        for k in range(len(self.picked)):
            if self.picked[k] is not None:
                new_pos[self.picked[k], :3] = pos[self.picked[k], :3] + deltas[k]
                new_pos[self.picked[k], 3] = 0.0
        st[:, 3:6] = st[:, 0:3]
        st[:, 0:3] = st[:, 0:3] + deltas
        sim.set_shape_states(st.astype(np.float32).ravel())
        sim.set_positions(new_pos.astype(np.float32).ravel())

    def movep(self, targets, speed, max_steps=1000, record=None):
        """simEnv.py:739-769: move both pickers toward targets by `speed` per sim step."""
        targets = np.asarray(targets, np.float64).reshape(-1, 3)
        for _ in range(max_steps):
            cur = self.picker_pos().astype(np.float64)
            delta = targets - cur
            dist = np.linalg.norm(delta, axis=1)
            if np.all(dist < 1e-4):
                return True
            step = np.where((dist > speed)[:, None], delta / np.maximum(dist, 1e-12)[:, None] * speed, delta)
            self.move(step)
            self.sim.step()
            if record is not None:
                record(self.sim)
        return False


def scenario_drop(sim, dimx=32, dimz=32, height=0.1, steps=60, record=None):
    """C1-style plumbing: pristine grid dropped from `height` onto the ground plane."""
    sim.set_scene(cloth_params(dimx, dimz, pos=(0.0, -height, 0.0)))
    for _ in range(steps):
        sim.step()
        if record is not None:
            record(sim)


def scenario_crumple(sim, dimx=32, dimz=32, seed=0, lift_steps=40, settle_steps=80, record=None):
    """Mirror of the task generator's crumple (environment/tasks.py:177-224): pin one particle, raise it, release."""
    rng = np.random.RandomState(seed)
    sim.set_scene(cloth_params(dimx, dimz, pos=(0.0, -0.2, 0.0)))
    sim.step()  # flex_utils.set_scene steps once before set_state (flex_utils.py:352)
    n = dimx * dimz
    w = sim.get_positions().reshape(-1, 4)[0, 3]
    p = flat_positions(dimx, dimz, y=0.0125, inv_mass=w)
    sim.set_positions(p.ravel())
    sim.set_velocities(np.zeros(3 * n, np.float32))
    k = int(rng.randint(n))
    target_h = 0.15 + 0.15 * rng.rand()
    pos = sim.get_positions().reshape(-1, 4).copy()
    saved_w = pos[k, 3]
    for s in range(lift_steps):
        pos = sim.get_positions().reshape(-1, 4).copy()
        pos[k, 1] += target_h / lift_steps
        pos[k, 3] = 0.0
        sim.set_positions(pos.ravel())
        sim.step()
        if record is not None:
            record(sim)
    pos = sim.get_positions().reshape(-1, 4).copy()
    pos[k, 3] = saved_w
    sim.set_positions(pos.ravel())
    for s in range(settle_steps):
        sim.step()
        if record is not None:
            record(sim)
    return k


def scenario_fling(sim, dimx=32, dimz=32, lift=0.25, fling_dist=0.15, settle_steps=40, record=None,
                   lift_speed=5e-3, fling_speed=6e-3):
    """Scripted two-corner fling (environment/simEnv.py:262-318 speeds): grasp two corners, lift, forward, back,
    lower, release, settle.  Exercises pinned particles, sphere contact, high velocity, ground friction and
    self-collision."""
    sim.set_scene(cloth_params(dimx, dimz, pos=(0.0, -0.2, 0.0)))
    sim.step()
    n = dimx * dimz
    w = sim.get_positions().reshape(-1, 4)[0, 3]
    p = flat_positions(dimx, dimz, y=0.0125, inv_mass=w)
    sim.set_positions(p.ravel())
    sim.set_velocities(np.zeros(3 * n, np.float32))
    c0, c1 = p[0, :3].astype(np.float64), p[dimx - 1, :3].astype(np.float64)
    pick = ScriptedPickers(sim, positions=(c0 + [0, 0.02, 0], c1 + [0, 0.02, 0]))
    pick.grasp_nearest(0)
    pick.grasp_nearest(1)
    assert pick.picked[0] is not None and pick.picked[1] is not None
    base = pick.picker_pos().astype(np.float64)
    up = base.copy(); up[:, 1] = lift
    pick.movep(up, lift_speed, record=record)
    fwd = up.copy(); fwd[:, 2] += fling_dist
    pick.movep(fwd, fling_speed, record=record)
    back = up.copy(); back[:, 2] -= fling_dist
    pick.movep(back, fling_speed, record=record)
    low = back.copy(); low[:, 1] = 0.05
    pick.movep(low, fling_speed, record=record)
    pick.release(0); pick.release(1)
    pick.move(np.zeros((2, 3)))
    for _ in range(settle_steps):
        sim.step()
        if record is not None:
            record(sim)
    return pick
