"""Scripted episodes used by the parity tests, the smoke test and bench.py.

Every scenario drives a duck-typed `sim` that exposes the reference's pyflex call names (set_scene, step,
get/set_positions, get/set_velocities, add_sphere, get/set_shape_states) -- either oracle.OracleSim or
flingbot_amd.sim.EnvView -- with identical fp32 inputs, so the two trajectories can be compared bit for bit.

The picker logic restates what the reference's callers do around the solver (environment/flex_utils.py:104-205
Picker._get_pos/_set_pos/step and environment/simEnv.py:739-769 movep): a kinematic sphere carries a pinned
(invMass = 0) particle by teleporting both by the same delta before each pyflex.step().
"""
import numpy as np


def cloth_params(dimx, dimz, pos=(0.0, -0.1, 0.0), stiff=(0.9, 0.9, 0.9), mass=0.5, flip=0):
    """scene_params[19] in the layout of flex_utils.py:332-342.  (Lives here, not in conftest.py, so that this module needs
    numpy only: tests/golden/capture_pyflex.py imports it on a machine that has PyFleX and no pytest.)"""
    return np.array([pos[0], pos[1], pos[2], dimx, dimz, stiff[0], stiff[1], stiff[2], 2,
                     0, 2, 0, np.pi / 2, -np.pi / 2, 0, 720, 720, mass, flip], dtype=np.float64)


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


# ---------------------------------------------------------------- SURVEY.md 8(d) canonical workloads C1 / C2 / scripted fling
# The same three recipes drive the parity tests, PARITY.md's sensitivity table (tests/parity_table.py), bench.py's C2 entry
# and -- on a machine with real PyFleX -- tests/golden/capture_pyflex.py, so a recorded PyFleX trajectory can be replayed
# here input for input.


def survey_params(dim):
    """scene_params = [0,1,0, dim,dim, .9,.9,.9, 2, 0,2,0, pi/2,-pi/2,0, 720,720, 0.5, 0]."""
    return cloth_params(dim, dim, pos=(0.0, 1.0, 0.0))


def set_to_flatten_positions(dimx, dimz, cloth_particle_radius=0.00625):
    """The array flex_utils.set_to_flatten (flex_utils.py:398-415) hands to pyflex.set_positions, expression by
    expression: linspace over dim * radius (so the pitch is dim / (dim - 1) * radius), y = radius, w = 1, then the
    mean of x, y, z subtracted -- which puts the sheet at y = 0 (to 1e-16), inside the ground's collision distance."""
    px = np.linspace(0, dimx * cloth_particle_radius, dimx)
    py = np.linspace(0, dimz * cloth_particle_radius, dimz)
    xx, yy = np.meshgrid(px, py)
    new_pos = np.empty(shape=(dimx * dimz, 4), dtype=np.float64)
    new_pos[:, 0] = xx.flatten()
    new_pos[:, 1] = cloth_particle_radius
    new_pos[:, 2] = yy.flatten()
    new_pos[:, 3] = 1.
    new_pos[:, :3] -= np.mean(new_pos[:, :3], axis=0)
    return new_pos


def canonical_flat(sim, dim):
    """flex_utils.set_scene (set_scene + its one step, flex_utils.py:343-354) followed by set_to_flatten: where every
    generated task starts (tasks.py:160-174)."""
    sim.set_scene(survey_params(dim))
    sim.step()
    sim.set_positions(set_to_flatten_positions(dim, dim).flatten())


def scenario_c1(sim, steps=200, record=None):
    """C1 = BASELINE.json configs[0]: 32 x 32, flattened, 200 pyflex.step() without rendering."""
    canonical_flat(sim, 32)
    for _ in range(steps):
        sim.step()
        if record is not None:
            record(sim)


def scenario_c2(sim, seed=0, dim=64, raise_steps=200, hold_steps=100, settle_steps=150, record=None):
    """C2 = the deterministic "hard task" crumple (mirror of tasks.py:177-224): flattened sheet, centre_object's step, pin
    particle seed % N, raise it to height 0.5 + u over 200 steps exactly like the generator's loop (position rewritten and
    velocity zeroed before every step), hold (fixed count instead of the generator's stability test, so that two solvers
    take the same number of steps), release, settle."""
    n = dim * dim
    canonical_flat(sim, dim)
    pos = sim.get_positions().reshape(-1, 4).copy()
    pos[:, [0, 2]] -= np.mean(pos[:, [0, 2]], axis=0, keepdims=True)       # center_object (flex_utils.py:310-314)
    sim.set_positions(pos.ravel())
    sim.step()
    if record is not None:
        record(sim)
    k = int(seed) % n
    height = float(np.random.RandomState(seed).random_sample(1)[0]) * 1.0 + 0.5
    cur = sim.get_positions().copy()
    w0 = cur[4 * k + 3]
    cur[4 * k + 3] = 0
    sim.set_positions(cur)
    pick = cur[4 * k: 4 * k + 3].copy()
    init_h = pick[1]
    speed = 1.0 / raise_steps

    def pinned_step():
        cur, vel = sim.get_positions().copy(), sim.get_velocities().copy()
        cur[4 * k: 4 * k + 3] = pick
        cur[4 * k + 3] = 0
        vel[3 * k: 3 * k + 3] = 0
        sim.set_positions(cur)
        sim.set_velocities(vel)
        sim.step()
        if record is not None:
            record(sim)

    for j in range(raise_steps):
        pick[1] = (height - init_h) * (j * speed) + init_h
        pinned_step()
    for _ in range(hold_steps):
        pinned_step()
    cur = sim.get_positions().copy()
    cur[4 * k + 3] = w0
    sim.set_positions(cur)
    for _ in range(settle_steps):
        sim.step()
        if record is not None:
            record(sim)
    return k


def scenario_c2_fling(sim, dim=64, settle_steps=300, record=None):
    """The scripted fling of SURVEY 8(d): pickers grasp the corners 0 and dim - 1, lift to y = 0.3 at 5e-3 per step, forward /
    back +-0.2 at 6e-3 per step (simEnv.py:262-275 speeds), lower, release, 300 settle steps."""
    return scenario_fling(sim, dim, dim, lift=0.3, fling_dist=0.2, settle_steps=settle_steps, record=record)


CANONICAL = {"c1": scenario_c1, "c2": scenario_c2, "fling": scenario_c2_fling}


class Recorder:
    """record= callback: keeps positions / velocities / shape states of frames 1, 10, 100, every `every`-th and (close()) the last."""

    def __init__(self, every=10, also=(1, 10, 100)):
        self.every, self.also, self.count = int(every), set(also), 0
        self.frames, self.pos, self.vel, self.shapes = [], [], [], []
        self._last = None

    def _snap(self, sim):
        sh = np.array(sim.get_shape_states(), np.float32).ravel()
        return (np.array(sim.get_positions(), np.float32).copy(), np.array(sim.get_velocities(), np.float32).copy(), sh)

    def __call__(self, sim):
        self.count += 1
        self._sim = sim
        if self.count % self.every == 0 or self.count in self.also:
            self._keep(self._snap(sim))

    def _keep(self, snap):
        if self.frames and self.frames[-1] == self.count:
            return
        self.frames.append(self.count)
        self.pos.append(snap[0]); self.vel.append(snap[1]); self.shapes.append(snap[2])

    def close(self):
        if self.count and (not self.frames or self.frames[-1] != self.count):
            self._keep(self._snap(self._sim))
        return self

    def arrays(self, prefix):
        ns = max((s.size for s in self.shapes), default=0)
        sh = np.zeros((len(self.frames), ns), np.float32)
        for i, s in enumerate(self.shapes):
            sh[i, :s.size] = s
        return {prefix + "frames": np.array(self.frames, np.int32), prefix + "positions": np.stack(self.pos),
                prefix + "velocities": np.stack(self.vel), prefix + "shape_states": sh}
