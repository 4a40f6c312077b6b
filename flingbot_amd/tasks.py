"""Batched task generation on the device (the generator half; HDF5 storage needs h5py, absent here).

Mirror of the reference's generate_randomization(task_difficulty='hard', grid cloth) -- environment/tasks.py:105-275 with
flex_utils.set_scene :320-355, set_to_flatten :398-415, center_object :313-317, Picker.reset :74-101 -- for MANY episodes
at once: every episode is an independent cloth that is flattened, grabbed at a random particle, lifted to a random height
over 200 steps, held until it hangs still, released and left to settle.  The reference moves the one pinned particle by
reading and rewriting the whole position / velocity arrays through pyflex every step (~500 steps per task); here the pin
is `fs_set_particles` (one particle per episode), the steps are batched launches, and the hold / settle loops read device
reductions.  Random draws are made with the reference's calls in the reference's order (numpy's and Python's global
generators), so seeding both reproduces the tasks the reference would generate on the same solver.

`sim` is a FlingSim (or anything with the same methods); no CPU fallback exists for the simulation itself.
"""
import random
from copy import deepcopy

import numpy as np


# ---- flex_utils.py:255-268, 358-395 (host numpy: used for the flattened sheet's area, which the reference computes from
#      the float64 array it is about to upload, not from simulator state)
def _vrange(start, end):
    n = int(np.max(end - start)) + 1
    return np.floor(np.arange(n) * (end - start)[:, None] / n + start[:, None]).astype("int")


def _vmeshgrid(vx, vy):
    n, k, d = vx.shape[0], vx.shape[1], vy.shape[1]
    vx = np.tile(vx[:, None, :], [1, d, 1]).reshape(n, -1)
    vy = np.tile(vy[:, :, None], [1, 1, k]).reshape(n, -1)
    return vx, vy


def covered_area_of(pos, cloth_particle_radius=0.00625):
    pos = np.reshape(pos, [-1, 4])
    min_x, min_y = np.min(pos[:, 0]), np.min(pos[:, 2])
    max_x, max_y = np.max(pos[:, 0]), np.max(pos[:, 2])
    init = np.array([min_x, min_y])
    span = np.array([max_x - min_x, max_y - min_y]) / 100.
    offset = pos[:, [0, 2]] - init
    x_lo = np.maximum(np.round((offset[:, 0] - cloth_particle_radius) / span[0]).astype(int), 0)
    x_hi = np.minimum(np.round((offset[:, 0] + cloth_particle_radius) / span[0]).astype(int), 100)
    y_lo = np.maximum(np.round((offset[:, 1] - cloth_particle_radius) / span[1]).astype(int), 0)
    y_hi = np.minimum(np.round((offset[:, 1] + cloth_particle_radius) / span[1]).astype(int), 100)
    grid = np.zeros(10000)
    xx, yy = _vmeshgrid(_vrange(x_lo, x_hi), _vrange(y_lo, y_hi))
    grid[np.clip((xx * 100 + yy).flatten(), 0, 9999)] = 1
    return np.sum(grid) * span[0] * span[1]


def flattened_positions(cloth_dimx, cloth_dimz, cloth_particle_radius=0.00625):
    """set_to_flatten (flex_utils.py:398-415): float64 [N,4], inverse mass 1."""
    n = cloth_dimx * cloth_dimz
    px = np.linspace(0, cloth_dimx * cloth_particle_radius, cloth_dimx)
    py = np.linspace(0, cloth_dimz * cloth_particle_radius, cloth_dimz)
    xx, yy = np.meshgrid(px, py)
    new_pos = np.empty(shape=(n, 4), dtype=float)
    new_pos[:, 0] = xx.flatten()
    new_pos[:, 1] = cloth_particle_radius
    new_pos[:, 2] = yy.flatten()
    new_pos[:, 3] = 1.
    new_pos[:, :3] -= np.mean(new_pos[:, :3], axis=0)
    return new_pos


def draw_task_parameters(min_cloth_size=64, strict_min_edge_length=64, max_cloth_size=104, task_difficulty='hard'):
    """The random draws of ONE generate_randomization call, in its order (tasks.py:120-122, 147-148; hard :179, 188; easy
    :229, 239 ten times), from the global numpy / Python generators.  Returns None where the reference returns None (both
    edges too short)."""
    cloth_dimx = np.random.randint(min_cloth_size, max_cloth_size)
    cloth_dimy = np.random.randint(min_cloth_size, max_cloth_size)
    if cloth_dimx < strict_min_edge_length and cloth_dimy < strict_min_edge_length:
        return None
    stiffness = np.random.uniform(0.85, 0.95, 3)
    cloth_mass = np.random.uniform(0.2, 2.0)
    out = dict(cloth_size=[cloth_dimx, cloth_dimy], cloth_stiff=stiffness, cloth_mass=cloth_mass,
               task_difficulty=task_difficulty)
    if task_difficulty == 'hard':
        out['pickpoint'] = random.randint(0, cloth_dimx * cloth_dimy - 1)
        out['height'] = np.random.random(1) * 1.0 + 0.5
    elif task_difficulty == 'easy':
        out['throws'] = []
        for _ in range(10):
            pickpoint = random.randint(0, cloth_dimx * cloth_dimy - 1)
            displacement = np.random.uniform(-0.2, 0.2, 3)
            displacement[1] = 0.2
            out['throws'].append((pickpoint, displacement))
    else:
        raise NotImplementedError()
    return out


def _picker_reset_states(center, picker_radius=0.05, num_picker=2):
    """Shape states after Picker.reset(center) (flex_utils.py:64-101)."""
    r = np.sqrt(num_picker - 1) * picker_radius * 2.
    pos = [[center[0] + np.cos(2 * np.pi * i / num_picker) * r, center[1], center[2] + np.sin(2 * np.pi * i / num_picker) * r]
           for i in range(num_picker)]
    return np.array(pos), np.array([np.hstack([p, p, [1, 0, 0, 0], [1, 0, 0, 0]]) for p in pos])


def _center_object(sim, envs):
    """center_object (flex_utils.py:313-317), then one step for those episodes."""
    for e in envs:
        pos = sim.get_positions(e).reshape(-1, 4)
        pos[:, [0, 2]] -= np.mean(pos[:, [0, 2]], axis=0, keepdims=True)
        sim.set_positions(e, pos.flatten())
    sim.step_list(envs, 1)


def generate_hard_tasks(sim, params, picker_radius=0.05):
    """Kept name of generate_tasks (both difficulties are handled there)."""
    return generate_tasks(sim, params, picker_radius)


def generate_tasks(sim, params, picker_radius=0.05):
    """params: one dict per episode of `sim` (draw_task_parameters(); None entries are skipped like the reference's `return
    None`), all of the same task_difficulty.  Returns one task dict per entry (None for skipped / rejected ones) with the
    reference's keys."""
    envs = [e for e, p in enumerate(params) if p is not None]
    tasks = [None] * len(params)
    if not envs:
        return tasks
    cam_pos, cam_angle = np.array([0, 2, 0]), np.array([np.pi * 0.5, -np.pi * 0.5, 0])
    flat_area = {}
    for e in envs:  # config + set_scene (flex_utils.py:320-355); the scene step follows for all of them at once
        p = params[e]
        scene_params = np.array([0, 1, 0, *p["cloth_size"], *p["cloth_stiff"], 2, *cam_pos, *cam_angle, 720, 720,
                                 p["cloth_mass"], 0])
        sim.set_scene(e, scene_params)
    sim.step_list(envs, 1)
    for e in envs:  # action_tool.reset([0, -1, 0]) and set_to_flatten
        centres, states = _picker_reset_states([0., -1., 0.], picker_radius)
        for c in centres:
            sim.add_sphere(e, picker_radius, c, [1, 0, 0, 0])
        sim.set_shape_states(e, sim.get_shape_states(e))
        sim.set_shape_states(e, states)
        new_pos = flattened_positions(*params[e]["cloth_size"])
        sim.set_positions(e, new_pos.flatten())
        flat_area[e] = covered_area_of(new_pos)
    _center_object(sim, envs)
    difficulty = params[envs[0]].get('task_difficulty', 'hard')
    assert all(params[e].get('task_difficulty', 'hard') == difficulty for e in envs), "one difficulty per batch"
    if difficulty == 'easy':
        _easy_throws(sim, envs, params)
    else:
        _hard_lift(sim, envs, params)
    return _finish_tasks(sim, envs, params, tasks, flat_area, difficulty)


def _easy_throws(sim, envs, params):
    """tasks.py:225-258: ten times, a random vertex is pinned and dragged 100 steps along a random displacement."""
    speed = 0.01
    for throw in range(10):
        pick = {e: int(params[e]['throws'][throw][0]) for e in envs}
        orig_w, start, target = {}, {}, {}
        for e in envs:
            curr = sim.get_positions(e)
            orig_w[e] = curr[pick[e] * 4 + 3]
            start[e] = curr[pick[e] * 4: pick[e] * 4 + 3].copy()            # float32
            target[e] = start[e] + params[e]['throws'][throw][1]             # float64
        sim.set_particles(envs, [pick[e] for e in envs], [[*start[e], 0.0] for e in envs], zero_velocity=False)
        for j in range(int(1 / speed)):
            at = {e: ((target[e] - start[e]) * (j * speed) + start[e]).astype(np.float32) for e in envs}
            sim.set_particles(envs, [pick[e] for e in envs], [[*at[e], 0.0] for e in envs], zero_velocity=True)
            sim.step_list(envs, 1)
        # reset to previous cloth parameters: the particle keeps where the solver has it (kinematic: the last pinned position)
        last = {e: ((target[e] - start[e]) * ((int(1 / speed) - 1) * speed) + start[e]).astype(np.float32) for e in envs}
        sim.set_particles(envs, [pick[e] for e in envs], [[*last[e], orig_w[e]] for e in envs], zero_velocity=False)


def _hard_lift(sim, envs, params):
    # ---- hard task (tasks.py:177-224): pin a random particle, raise it over 200 steps ...
    pick = {e: int(params[e]["pickpoint"]) for e in envs}
    orig_w, pick_pos, init_h = {}, {}, {}
    for e in envs:
        curr = sim.get_positions(e)
        orig_w[e] = curr[pick[e] * 4 + 3]
        pick_pos[e] = curr[pick[e] * 4: pick[e] * 4 + 3].copy()  # float32
        init_h[e] = pick_pos[e][1]
    speed = 0.005

    def pin(es):
        sim.set_particles(es, [pick[e] for e in es], [[*pick_pos[e], 0.0] for e in es], zero_velocity=True)

    for j in range(int(1 / speed)):
        for e in envs:
            pick_pos[e][1] = ((params[e]["height"] - init_h[e]) * (j * speed) + init_h[e])[0]  # float64 -> float32 element
        pin(envs)
        sim.step_list(envs, 1)
    # ... hold it until the cloth hangs still (wait_until_stable(max_steps=1, tolerance=1e-1) steps once more when it is not)
    active = list(envs)
    for j in range(0, 300):
        if not active:
            break
        pin(active)
        sim.step_list(active, 1)
        vmax = sim.cloth_stats(active)[:, 2]
        moving = [e for e, v in zip(active, vmax) if not (v < 1e-1)]
        if moving:
            sim.step_list(moving, 1)
        active = [e for e, v in zip(active, vmax) if not ((v < 1e-1) and j > 5)]
    # release: the pinned particle gets its inverse mass back (tasks.py:221-224)
    sim.set_particles(envs, [pick[e] for e in envs], [[*pick_pos[e], orig_w[e]] for e in envs], zero_velocity=False)


def _finish_tasks(sim, envs, params, tasks, flat_area, difficulty):
    sim.wait_until_stable(envs)  # wait_until_stable(gui=gui): 300 steps at most, tolerance 1e-2
    stats = sim.cloth_stats(envs)
    keep = [e for e, s_ in zip(envs, stats) if not (s_[1] > 0.4)]  # heights.max() > 0.4: "probably an error" -> None
    _center_object(sim, keep)
    coverage = sim.coverage()
    for e in keep:
        p = params[e]
        tasks[e] = {
            'particle_pos': sim.get_positions(e), 'particle_vel': sim.get_velocities(e), 'initial_coverage': coverage[e],
            'shape_pos': sim.get_shape_states(e), 'phase': sim.get_phases(e), 'flatten_area': flat_area[e], 'flip_mesh': 0,
            'cloth_size': np.array(p["cloth_size"]), 'cloth_stiff': p["cloth_stiff"], 'cloth_mass': p["cloth_mass"],
            'task_difficulty': difficulty, 'mesh_verts': np.array([]), 'mesh_stretch_edges': np.array([]),
            'mesh_bend_edges': np.array([]), 'mesh_shear_edges': np.array([]), 'mesh_faces': np.array([]),
        }
    return tasks


def load_tasks(sim, tasks, cloth_pos=(0, 2, 0)):
    """set_scene(config=task.get_config(), state=task.get_state()) (flex_utils.py:320-355, tasks.py:374-411) for episode e =
    index in `tasks`: build the scene, step once, then overwrite positions / velocities / phases (and the shape states when
    the scene has shapes: with none, pyflex.set_shape_states copies nothing).  None entries are left untouched."""
    envs = [e for e, t in enumerate(tasks) if t is not None]
    cam_pos, cam_angle = np.array([0, 2, 0]), np.array([np.pi * 0.5, -np.pi * 0.5, 0])
    for e in envs:
        t = tasks[e]
        size = [-1, -1] if len(t['mesh_verts']) > 0 else list(t['cloth_size'])
        scene_params = np.array([*cloth_pos, *size, *t['cloth_stiff'], 2, *cam_pos, *cam_angle, 720, 720, t['cloth_mass'],
                                 t['flip_mesh']])
        sim.set_scene(e, scene_params, t['mesh_verts'], t['mesh_stretch_edges'], t['mesh_bend_edges'],
                      t['mesh_shear_edges'], t['mesh_faces'])
    sim.step_list(envs, 1)
    for e in envs:
        t = tasks[e]
        sim.set_positions(e, t['particle_pos'])
        sim.set_velocities(e, t['particle_vel'])
        if sim.n_shapes(e) > 0:
            sim.set_shape_states(e, t['shape_pos'])
        sim.set_phases(e, t['phase'])
        sim.set_camera_params(e, [*cam_pos, *cam_angle, 720, 720])
    return envs
