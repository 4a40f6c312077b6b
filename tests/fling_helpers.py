"""Shared by the fling-primitive tests: the golden cases and a CPU stand-in for FlingSim built on the oracle (test
infrastructure only), so `flingbot_amd.primitives.FlingPrimitives` -- pure host logic -- can be checked without a GPU."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _each(fn, items):
    """[fn(x) for x in items] with the calls on threads of their own when there is more than one: the oracle stand-ins below
    hold one INDEPENDENT CPU oracle per episode, orc_step runs without the GIL (ctypes), and the suite's wall time is mostly
    those steps.  Order of the results = order of the items; every call touches its own episode only."""
    items = list(items)
    if len(items) <= 1:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(items), os.cpu_count() or 1)) as pool:
        return list(pool.map(fn, items))


def picker_centres():
    """Picker.reset([0, 0.1, 0]) sphere centres (flex_utils.py:82-97)."""
    r = np.sqrt(2 - 1) * 0.02 * 2.
    return [[0.0 + np.cos(2 * np.pi * i / 2) * r, 0.1, 0.0 + np.sin(2 * np.pi * i / 2) * r] for i in range(2)]


def load_fling_golden():
    return np.load(os.path.join(GOLD, "fling_golden.npz"))


class OracleBatch:
    """The slice of the FlingSim interface FlingPrimitives uses, on N independent CPU oracles + the numpy restatement of
    the reference picker (oracle/picker.py).  The reductions are the reference's own numpy expressions."""

    def __init__(self, n, scene_params, init_pos, pickers=True, variant=None):
        from oracle import OracleSim
        from oracle.picker import OraclePicker

        self.sims, self.tools, self.snap = [], [], {}
        for _ in range(n):
            o = OracleSim(variant)   # variant: a sensitivity build of the oracle (tests/parity_table.py --plausibility)
            o.set_scene(scene_params)
            o.step(1)
            o.set_positions(init_pos.ravel())
            o.set_velocities(np.zeros(3 * init_pos.shape[0], np.float32))
            t = OraclePicker(o)
            if pickers:
                t.reset(picker_centres())
            self.sims.append(o)
            self.tools.append(t)

    # ---- what FlingPrimitives.setup_pickers / BatchedFlingEnv.step_actions need beyond the primitives
    def add_sphere(self, e, radius, pos, quat):
        self.sims[e].add_sphere(radius, pos, quat)

    def set_shape_states(self, e, s):
        self.sims[e].set_shape_states(s)

    def picker_reset(self, e, picker_threshold=0.005, particle_radius=0.00625, picker_radius=None):
        t = self.tools[e]
        t.picker_threshold, t.particle_radius = picker_threshold, particle_radius
        if picker_radius is not None:
            t.picker_radius = picker_radius
        t.picked_particles = [None] * t.num_picker
        t.particle_inv_mass = self.sims[e].get_positions().reshape(-1, 4)[:, 3]

    def step_list(self, envs, n_steps=1):
        _each(lambda e: self.sims[e].step(n_steps), envs)

    def coverage(self):
        from oracle.coverage import covered_area

        return [covered_area(s.get_positions()) for s in self.sims]

    def snapshot_positions(self, envs):
        self._touch(envs)
        for e in envs:
            self.snap[e] = self.sims[e].get_positions().reshape(-1, 4)[:, :3].copy()  # SimEnv.preaction

    def max_displacement(self, envs):
        self._touch(envs)
        out = []
        for e in envs:
            post = self.sims[e].get_positions().reshape(-1, 4)[:, :3]
            out.append(np.linalg.norm(np.abs(post - self.snap[e]), axis=1).max())  # simEnv.py:470-472
        return np.array(out, np.float32)

    def movep(self, envs, targets, grasp, speed=0.1, limit=1000, min_steps=None, eps=1e-4):
        targets = np.asarray(targets)
        iters = np.array(_each(lambda ke: self.tools[ke[1]].movep(targets[ke[0]], [bool(g) for g in grasp[ke[0]]], speed=speed,
                                                                 limit=limit, min_steps=min_steps, eps=eps),
                               list(enumerate(envs))), np.int32)
        self.last_movep_steps = sum(self.tools[e].last_sim_steps for e in envs)
        return iters

    def advance(self, envs, kind, targets, grasp, speed, limit, min_steps, f32, start, cap_min=8, cap=64, eps=1e-4,
                tolerance=1e-2):
        """FlingSim.advance (fs_advance) on the CPU oracles: every episode takes its next chunk of its own loop.  The chunk
        length follows fs_advance's rule (the shortest mover's remaining steps, at least cap_min, at most cap) so that the
        sequence of calls equals the device's; episodes are independent, so the rule cannot change any result."""
        n = len(envs)
        prog, status, steps = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.advance_calls = getattr(self, "advance_calls", 0) + 1
        chunk = cap if not any(k == 0 for k in kind) else max(cap_min, 1)  # CPU stand-in: fixed short chunks (worst case for resumption)
        tols = np.broadcast_to(np.asarray(tolerance, np.float64), (n,))
        def one(ae):
            a, e = ae
            if kind[a] == 0:
                tg = np.asarray(targets[a], np.float32 if f32[a] else np.float64).reshape(-1, 3)
                ms = None if min_steps[a] < 0 else int(min_steps[a])
                prog[a], status[a] = self.tools[e].movep(tg, [bool(g) for g in np.asarray(grasp[a]).reshape(-1)], speed=speed[a],
                                                         limit=int(limit[a]), min_steps=ms, eps=eps, start=int(start[a]),
                                                         max_sim_steps=chunk)
                steps[a] = self.tools[e].last_sim_steps
            else:
                done, st = 0, 0
                while True:
                    if start[a] + done >= limit[a]:
                        st = 2
                        break
                    if done >= chunk:
                        break
                    if kind[a] == 1 and np.abs(self.sims[e].get_velocities()).max() < tols[a]:
                        st = 1
                        break
                    self.sims[e].step(1)
                    done += 1
                prog[a], status[a], steps[a] = start[a] + done, (1 if st == 2 and kind[a] == 2 else st), done

        _each(one, list(enumerate(envs)))
        return prog, status, steps

    # ---- fs_advance_begin / fs_advance_end / fs_service_lane on the CPU oracles, with the PROTOCOL checked: the work of a
    # chunk is done at once here (an oracle has no queue), but a wait / step entry's outcome stays hidden until advance_end,
    # start = -1 continues from the loop state kept per episode, and every call that reads or writes an episode asserts that
    # the episode is not a live part of an open chunk and that the lane is the right one
    def _live(self, e):
        return any(e in t["live"] for t in getattr(self, "_open", {}).values())

    def _touch(self, envs):
        for e in envs:
            assert not self._live(int(e)), f"episode {e} touched while it is part of a chunk in flight"
        if getattr(self, "_open", None):
            assert getattr(self, "_lane", False), "host-side work while chunks are open must run on the service lane"

    def service_lane(self, on):
        self._lane = bool(on)

    def advance_timing(self):
        return dict(calls=getattr(self, "advance_calls", 0), sequences=getattr(self, "_sequences", 0), wall_ms=0.0, gpu_ms=0.0, prep_ms=0.0)

    def advance_in_flight(self):
        return len(getattr(self, "_open", {}))

    def advance_begin(self, envs, kind, targets, grasp, speed, limit, min_steps, f32, start, cap_min=8, cap=64, eps=1e-4,
                      tolerance=1e-2):
        assert not getattr(self, "_lane", False), "chunks are queued on the main lane"
        self._open = getattr(self, "_open", {})
        self._wait = getattr(self, "_wait", {})
        assert len(self._open) < 4
        n = len(envs)
        start = [int(x) for x in start]
        tols = np.broadcast_to(np.asarray(tolerance, np.float64), (n,))
        for a, e in enumerate(envs):  # loop states: set (start >= 0) or continued (start = -1)
            if kind[a] != 0 and start[a] >= 0:
                self._wait[int(e)] = dict(steps=start[a], stable=False, over=False)
            assert kind[a] != 0 or start[a] >= 0
        known = [start[a] if (kind[a] == 0 or start[a] >= 0) else self._wait[int(envs[a])]["steps"] for a in range(n)]
        gone = [kind[a] != 0 and start[a] < 0 and self._wait[int(envs[a])]["over"] for a in range(n)]
        act = [a for a in range(n) if not gone[a]]
        prog, status, steps = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        if act:
            p2, s2, t2 = self.advance([envs[a] for a in act], [kind[a] for a in act], [targets[a] for a in act],
                                      [grasp[a] for a in act], [speed[a] for a in act], [limit[a] for a in act],
                                      [min_steps[a] for a in act], [f32[a] for a in act], [known[a] for a in act],
                                      cap_min=cap_min, cap=cap, eps=eps, tolerance=[tols[a] for a in act])
            for k, a in enumerate(act):
                prog[a], status[a], steps[a] = p2[k], s2[k], t2[k]
        self._sequences = getattr(self, "_sequences", 0) + (int(steps.max()) if n else 0)
        hidden, live = {}, set()
        for a, e in enumerate(envs):
            e = int(e)
            if kind[a] == 0:
                live.add(e)
                continue
            if start[a] >= 0 and limit[a] - start[a] <= 0:
                continue  # answered at once, like the library
            w = self._wait[e]
            if not gone[a]:
                w["steps"] = int(prog[a])
                w["stable"] = bool(kind[a] == 1 and status[a] == 1)
                w["over"] = bool(status[a] != 0)
                live.add(e)  # (the host cannot know yet that the loop may have ended)
            hidden[a] = (w["steps"], 1 if w["stable"] else ((1 if kind[a] == 2 else 2) if w["over"] else 0),
                         (w["steps"] - start[a]) if start[a] >= 0 else -1)
            prog[a], status[a], steps[a] = max(start[a], 0), -1, 0
        ticket = max(self._open, default=-1) + 1
        self._open[ticket] = dict(hidden=hidden, live=live)
        return ticket, prog, status, steps

    def advance_end(self, ticket, prog, status, steps):
        t = self._open.pop(ticket)
        for a, (p, s, d) in t["hidden"].items():
            prog[a], status[a], steps[a] = p, s, d
        return prog, status, steps

    def get_shape_states(self, e):
        return self.sims[e].get_shape_states()

    def get_positions(self, e):
        self._touch([e])
        return self.sims[e].get_positions()

    def cloth_stats(self, envs):
        self._touch(envs)
        out = np.empty((len(envs), 3), np.float32)
        for k, e in enumerate(envs):
            pos = self.sims[e].get_positions().reshape(-1, 4)
            out[k] = (pos[:, 1].min(), pos[:, 1].max(), np.abs(self.sims[e].get_velocities()).max())
        return out

    def stretch_probe(self, envs, midpoints_xz, height_thr):
        self._touch(envs)
        single, near = [], []
        for k, e in enumerate(envs):
            positions = self.sims[e].get_positions().reshape((-1, 4))[:, :3]
            high_positions = positions[positions[:, 1] > height_thr[k], ...]
            single.append(bool((high_positions[:, 0] < 0).all() or (high_positions[:, 0] > 0).all()))
            mid = np.asarray(midpoints_xz[k], np.float32)
            plist = [p for p in positions]
            plist.sort(key=lambda pos: np.linalg.norm(pos[[0, 2]] - mid))  # simEnv.py:163-165
            near.append(plist[0])
        return np.array(single), np.array(near, np.float32)

    def wait_until_stable(self, envs, max_steps=300, tolerance=1e-2):
        def one(e):
            done, ok = 0, False
            for _ in range(max_steps):  # flex_utils.py:430-441
                if np.abs(self.sims[e].get_velocities()).max() < tolerance:
                    ok = True
                    break
                self.sims[e].step(1)
                done += 1
            return ok, done

        res = _each(one, envs)
        return np.array([r[0] for r in res]), np.array([r[1] for r in res], np.int32)


def load_primitives_golden():
    return np.load(os.path.join(GOLD, "primitives_golden.npz"))


def run_primitives_golden(make_sim, get_positions, get_shapes, scheduled=False):
    """Runs the golden cases of the other manipulation primitives (drag / place / stretchdrag) on a simulator made by
    make_sim(n) and checks final particle positions and picker states bit for bit.  scheduled=False: batched per kind
    through the lock-step primitives; True: ALL cases of all kinds at once as per-episode programs (schedule.py)."""
    from flingbot_amd.primitives import FlingPrimitives

    g = load_primitives_golden()
    kinds = [str(k) for k in g["kind"]]
    groups = [list(range(len(kinds)))] if scheduled else [[c for c, k in enumerate(kinds) if k == kind]
                                                           for kind in ("drag", "place", "stretchdrag")]
    for cases in groups:
        kind = kinds[cases[0]]
        sim = make_sim(len(cases))
        prim = FlingPrimitives(sim, range(len(cases)), stretchdrag_dist=float(g["stretchdrag_dist"]))
        if scheduled:
            res, _ = prim.act_scheduled({k: (kinds[c], g["p1"][c], g["p2"][c], g["g1"][c], g["g2"][c])
                                         for k, c in enumerate(cases)}, settle=False, cap_min=3, cap=7)
            out = [res[k] for k in range(len(cases))]
        else:
            fn = {"drag": prim.pick_and_drag, "place": prim.pick_and_place, "stretchdrag": prim.pick_stretch_drag}[kind]
            out = fn(g["p1"][cases], g["p2"][cases], g["g1"][cases], g["g2"][cases])
        for k, c in enumerate(cases):
            kind = kinds[c]
            assert out[k]["skipped"] == (g["steps"][c] == 0), (kind, c)
            if kind == "stretchdrag" and not np.isnan(g["stretch_ret"][c]):
                assert out[k]["dist"] == g["stretch_ret"][c], (kind, c)
            assert np.array_equal(get_positions(sim, k).view(np.uint32), g["pos"][c].view(np.uint32)), (kind, c)
            assert np.array_equal(np.asarray(get_shapes(sim, k), np.float32).view(np.uint32), g["shapes"][c].view(np.uint32)), (kind, c)


class OracleTaskSim:
    """The slice of the FlingSim interface flingbot_amd.tasks uses, on N bare CPU oracles (no scene yet)."""

    def __init__(self, n, variant=None):
        from oracle import OracleSim

        self.sims = [OracleSim(variant) for _ in range(n)]

    def set_scene(self, e, scene_params):
        self.sims[e].set_scene(scene_params)

    def step_list(self, envs, n_steps=1):
        _each(lambda e: self.sims[e].step(n_steps), envs)

    def add_sphere(self, e, radius, pos, quat):
        self.sims[e].add_sphere(radius, pos, quat)

    def get_shape_states(self, e):
        return self.sims[e].get_shape_states()

    def set_shape_states(self, e, s):
        self.sims[e].set_shape_states(s)

    def get_positions(self, e):
        return self.sims[e].get_positions()

    def set_positions(self, e, p):
        self.sims[e].set_positions(p)

    def get_velocities(self, e):
        return self.sims[e].get_velocities()

    def get_phases(self, e):
        return self.sims[e].get_phases()

    def set_particles(self, envs, pids, pos4, zero_velocity=True):
        for e, pid, p4 in zip(envs, pids, pos4):
            pos = self.sims[e].get_positions().reshape(-1, 4).copy()
            pos[pid] = np.asarray(p4, np.float32)
            self.sims[e].set_positions(pos.ravel())
            if zero_velocity:
                vel = self.sims[e].get_velocities().reshape(-1, 3).copy()
                vel[pid] = 0
                self.sims[e].set_velocities(vel.ravel())

    def cloth_stats(self, envs):
        out = np.empty((len(envs), 3), np.float32)
        for k, e in enumerate(envs):
            pos = self.sims[e].get_positions().reshape(-1, 4)
            out[k] = (pos[:, 1].min(), pos[:, 1].max(), np.abs(self.sims[e].get_velocities()).max())
        return out

    def wait_until_stable(self, envs, max_steps=300, tolerance=1e-2):
        def one(e):
            done, ok = 0, False
            for _ in range(max_steps):
                if np.abs(self.sims[e].get_velocities()).max() < tolerance:
                    ok = True
                    break
                self.sims[e].step(1)
                done += 1
            return ok, done

        res = _each(one, envs)
        return np.array([r[0] for r in res]), np.array([r[1] for r in res], np.int32)

    def coverage(self):
        from oracle.coverage import covered_area

        return [covered_area(s.get_positions()) if s.n else 0.0 for s in self.sims]


_TASKS_ON_ORACLE = {}


def oracle_generated_tasks():
    """The golden tasks regenerated on the CPU oracle (checked against the reference's), made once per test session."""
    if "tasks" not in _TASKS_ON_ORACLE:
        _TASKS_ON_ORACLE["tasks"] = check_tasks_against_golden(lambda n: OracleTaskSim(n))
    return _TASKS_ON_ORACLE["tasks"]


def check_tasks_against_golden(make_sim):
    """Seeds numpy's and Python's generators like the golden run, draws the task parameters with the product's
    draw_task_parameters and generates the tasks on make_sim(n); everything the reference returned must be reproduced."""
    import random

    from flingbot_amd import tasks as ftasks

    g = np.load(os.path.join(GOLD, "task_golden.npz"))
    n_cases = sum(1 for k in g.files if k.endswith("_seed"))
    results = {}
    for difficulty in ("hard", "easy"):
        cases = [ci for ci in range(n_cases) if str(g[f"t{ci}_difficulty"]) == difficulty]
        params = []
        for ci in cases:
            seed = int(g[f"t{ci}_seed"])
            random.seed(seed)
            np.random.seed(seed)
            params.append(ftasks.draw_task_parameters(min_cloth_size=20, strict_min_edge_length=20, max_cloth_size=30,
                                                      task_difficulty=difficulty))
        sim = make_sim(len(cases))
        out = ftasks.generate_tasks(sim, params)
        for ci, task in zip(cases, out):
            assert task is not None and task["task_difficulty"] == difficulty
            assert task["cloth_size"].tolist() == g[f"t{ci}_cloth_size"].tolist()
            assert np.array_equal(task["cloth_stiff"], g[f"t{ci}_cloth_stiff"]) and task["cloth_mass"] == float(g[f"t{ci}_cloth_mass"])
            assert task["flatten_area"] == float(g[f"t{ci}_flatten_area"])
            for k in ("particle_pos", "particle_vel", "shape_pos"):
                assert np.array_equal(np.asarray(task[k], np.float32).view(np.uint32), g[f"t{ci}_{k}"].view(np.uint32)), (ci, k)
            assert np.array_equal(task["phase"], g[f"t{ci}_phase"])
            assert abs(task["initial_coverage"] - float(g[f"t{ci}_initial_coverage"])) <= 1e-12
            results[ci] = task
    return results


def load_step_golden():
    return np.load(os.path.join(GOLD, "step_golden.npz"))


def run_step_golden(make_sim, get_positions, get_shapes, scheduled=False):
    """SimEnv.step's bookkeeping (tests/golden/step_golden.npz, recorded from the reference's own SimEnv.step with the action
    selection scripted) through BatchedFlingEnv.step_actions on a simulator made by make_sim(n): every case is its own
    episode, all advanced together step by step; rewards, termination, timesteps, simulation-step counts, grasp flags,
    particle positions and picker states must be reproduced exactly."""
    from flingbot_amd.env import BatchedFlingEnv

    g = load_step_golden()
    n = int(g["n_cases"])
    sim = make_sim(n)
    env = BatchedFlingEnv.__new__(BatchedFlingEnv)  # host bookkeeping only: no selector / CUDA pieces are touched
    env.sim = sim
    env.actions = ["fling", "stretchdrag", "drag", "place"]
    env._prim_kwargs = dict(grasp_height=0.02, fling_speed=6e-3, stretchdrag_dist=0.3)
    env.episode_length = 0
    env.scheduled = scheduled  # True: per-episode programs on shared launch sequences instead of the lock-step phases
    env.attach(range(n))
    lengths = {c: int(g[f"c{c}_episode_length"]) for c in range(n)}
    for c in range(n):
        assert abs(env.init_coverage[c] - float(g[f"c{c}_init_coverage"])) <= 1e-15
    max_steps = max(len(g[f"c{c}_terminate"]) for c in range(n))
    for k in range(max_steps):
        run = [c for c in range(n) if k < len(g[f"c{c}_terminate"])]
        chosen = {}
        for c in run:
            prim = str(g[f"c{c}_prim"][k])
            if prim != "None":
                chosen[c] = (prim, dict(p1=g[f"c{c}_p1"][k].copy(), p2=g[f"c{c}_p2"][k].copy(),
                                        p1_grasp_cloth=bool(g[f"c{c}_g1"][k]), p2_grasp_cloth=bool(g[f"c{c}_g2"][k])))
        before = env.prim.sim_steps
        # episode_length differs per case: evaluate the termination rule per episode
        rewards, acted = {}, {}
        for length in sorted(set(lengths[c] for c in run)):
            sub = [c for c in run if lengths[c] == length]
            env.episode_length = length
            r, a = env.step_actions(sub, chosen)
            rewards.update(r)
            acted.update(a)
        for c in run:
            assert rewards[c] == float(g[f"c{c}_reward"][k]), (c, k, rewards[c], float(g[f"c{c}_reward"][k]))
            assert env.terminate[c] == bool(g[f"c{c}_terminate"][k]), (c, k)
            assert env.timestep[c] == int(g[f"c{c}_timestep"][k]), (c, k)
            assert [bool(x) for x in env.prim.grasp_states[c]] == [bool(x) for x in g[f"c{c}_grasp"][k]], (c, k)
            assert np.array_equal(get_positions(sim, c).view(np.uint32), g[f"c{c}_pos"][k].view(np.uint32)), (c, k)
            assert np.array_equal(np.asarray(get_shapes(sim, c), np.float32).view(np.uint32),
                                  g[f"c{c}_shapes"][k].view(np.uint32)), (c, k)
            assert acted[c] == (None if str(g[f"c{c}_prim"][k]) == "None" else str(g[f"c{c}_prim"][k]))
        assert env.prim.sim_steps - before == sum(int(g[f"c{c}_sim_steps"][k]) for c in run), k
    assert any(bool(g[f"c{c}_terminate"][-1]) and int(g[f"c{c}_timestep"][-1]) < lengths[c] for c in range(n))
    return env
