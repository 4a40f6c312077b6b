"""Batched fling primitive on the device-resident simulator.

Host-side mirror of the reference's manipulation code for MANY episodes at once -- same control flow, same numpy
expressions (and therefore the same float32 / float64 roundings) per episode:

    SimEnv.pick_and_fling_primitive   environment/simEnv.py:283-318
    SimEnv.pick_and_drag_primitive    environment/simEnv.py:320-345
    SimEnv.pick_and_place_primitive   environment/simEnv.py:347-374
    SimEnv.pick_stretch_drag_primitive  environment/simEnv.py:376-428
    SimEnv.stretch_cloth              environment/simEnv.py:140-184
    SimEnv.lift_cloth                 environment/simEnv.py:186-200
    SimEnv.fling_primitive            environment/simEnv.py:262-281
    SimEnv.is_cloth_grasped           environment/simEnv.py:809-813
    SimEnv.reset_end_effectors / set_grasp / movep   :739-780

Everything that touches particle data runs on the GPU: `movep` is `fs_movep_batch` (every simulation step of every
episode without a host round trip), the loops' tests read device reductions (`fs_cloth_stats`, `fs_stretch_probe`)
instead of downloading 64 KiB of positions per episode and trip.  Episodes are independent, so advancing them stage by
stage in lock step gives each one exactly the trajectory the reference's sequential code gives it.
"""
import numpy as np

from .sim import FlingSim


class FlingPrimitives:
    def __init__(self, sim: FlingSim, envs, grasp_height=0.02, fling_speed=6e-3, fixed_fling_height=-1,
                 stretchdrag_dist=0.3):
        self.sim = sim
        self.envs = np.asarray(envs, np.int32).reshape(-1)
        self.grasp_height = grasp_height
        self.fling_speed = fling_speed
        self.fixed_fling_height = fixed_fling_height
        self.stretchdrag_dist = stretchdrag_dist
        self.grasp_states = {int(e): [False, False] for e in self.envs}
        self.terminate = {int(e): False for e in self.envs}
        self.sim_steps = 0  # simulation steps issued by this object (all episodes)

    # ---- SimEnv.reset after set_scene (simEnv.py:674-681): action_tool.reset([0.2, 0.5, 0.0]), reset_end_effectors,
    #      one simulation step, set_grasp(False)
    def setup_pickers(self, center=(0.2, 0.5, 0.0), picker_radius=None, picker_threshold=0.005, particle_radius=0.00625):
        for e in (int(e) for e in self.envs):
            self.place_pickers(e, center, picker_radius, picker_threshold, particle_radius)
        self.reset_end_effectors(self.envs)
        self.sim.step_list([int(e) for e in self.envs], 1)
        self.sim_steps += len(self.envs)
        self.set_grasp(self.envs, False)

    def place_pickers(self, e, center=(0.2, 0.5, 0.0), picker_radius=None, picker_threshold=0.005, particle_radius=0.00625):
        """Picker.reset(center) for one episode (flex_utils.py:82-101): the two spheres, their states, the picker bookkeeping."""
        r_p = self.grasp_height if picker_radius is None else picker_radius  # picker_radius = grasp_height (simEnv.py:129-134)
        r = np.sqrt(2 - 1) * r_p * 2.
        centres = [[center[0] + np.cos(2 * np.pi * i / 2) * r, center[1], center[2] + np.sin(2 * np.pi * i / 2) * r]
                   for i in range(2)]
        for c in centres:
            self.sim.add_sphere(e, r_p, c, [1, 0, 0, 0])
        self.sim.set_shape_states(e, self.sim.get_shape_states(e))
        self.sim.set_shape_states(e, np.array([np.hstack([c, c, [1, 0, 0, 0], [1, 0, 0, 0]]) for c in centres]))
        self.sim.picker_reset(e, picker_threshold, particle_radius, picker_radius=r_p)

    # ---- SimEnv.movep for a subset of the episodes, each with its own targets
    def movep(self, envs, targets, speed=None, min_steps=None, limit=1000):
        envs = [int(e) for e in envs]
        if not envs:
            return
        speed = 0.1 if speed is None else speed  # dump_visualizations is off in batch mode (simEnv.py:740-744)
        grasp = [self.grasp_states[e] for e in envs]
        self.sim.movep(envs, np.array(targets), grasp, speed=speed, limit=limit, min_steps=min_steps)
        self.sim_steps += int(self.sim.last_movep_steps)  # iterations that find the pickers on target take no step

    def set_grasp(self, envs, grasp):
        for e in envs:
            self.grasp_states[int(e)] = [bool(grasp)] * 2 if isinstance(grasp, (bool, np.bool_)) else [bool(g) for g in grasp]

    def picker_positions(self, e):
        """Picker._get_pos()[0]: float32 [2,3] (flex_utils.py:104-112)."""
        return np.array(self.sim.get_shape_states(e)).reshape(-1, 14)[:, :3]

    def is_cloth_grasped(self, envs):
        st = self.sim.cloth_stats(envs)
        return st[:, 1] > 0.2  # heights.max() > 0.2, float32 vs python float as numpy compares them

    # ---- simEnv.py:140-184, one state machine per episode, advanced together
    def stretch_cloth(self, envs, grasp_dist, fling_height=0.7, max_grasp_dist=0.7, increment_step=0.02):
        envs = [int(e) for e in envs]
        result = {}
        st = {}
        first_targets = []
        for e, gd in zip(envs, grasp_dist):
            gd = np.float64(gd)
            left, right = self.picker_positions(e)
            left[1] = fling_height
            right[1] = fling_height
            midpoint = (left + right) / 2
            direction = left - right
            direction = direction / np.linalg.norm(direction)
            st[e] = dict(grasp_dist=gd, midpoint=midpoint, direction=direction, stable_steps=0, cloth_midpoint=1e2)
            first_targets.append([left, right])
        self.movep(envs, first_targets, speed=5e-4, min_steps=20)
        active = list(envs)
        while active:
            mids = [st[e]["midpoint"][[0, 2]] for e in active]
            thr = [np.float32(fling_height - 0.1)] * len(active)
            single, nearest = self.sim.stretch_probe(active, mids, thr)
            nxt, targets = [], []
            for k, e in enumerate(active):
                s = st[e]
                if single[k]:  # single grasp
                    result[e] = s["grasp_dist"]
                    continue
                new_cloth_midpoint = nearest[k]
                stable = np.linalg.norm(new_cloth_midpoint - s["cloth_midpoint"]) < 1.5e-2
                s["stable_steps"] = s["stable_steps"] + 1 if stable else 0
                if s["stable_steps"] > 2:
                    result[e] = s["grasp_dist"]
                    continue
                s["cloth_midpoint"] = new_cloth_midpoint
                s["grasp_dist"] += increment_step
                # float32 arrays meet the np.float64 scalar grasp_dist (np.linalg.norm's return type in every caller):
                # float64 under NumPy >= 2 (NEP 50), which is what the goldens pin -- spelled out so the targets do not
                # depend on the installed NumPy's promotion rules (NumPy 1.x would keep float32 here)
                mid64, dir64 = s["midpoint"].astype(np.float64), s["direction"].astype(np.float64)
                left = mid64 + dir64 * np.float64(s["grasp_dist"]) / 2
                right = mid64 - dir64 * np.float64(s["grasp_dist"]) / 2
                nxt.append(e)
                targets.append([left, right])
            self.movep(nxt, targets, speed=5e-4)
            active = []
            for e in nxt:
                if st[e]["grasp_dist"] > max_grasp_dist:
                    result[e] = max_grasp_dist
                else:
                    active.append(e)
        return [result[e] for e in envs]

    # ---- simEnv.py:186-200
    def lift_cloth(self, envs, grasp_dist, fling_height=0.7, increment_step=0.05, max_height=0.7):
        envs = [int(e) for e in envs]
        height = {e: fling_height for e in envs}
        dist = dict(zip(envs, grasp_dist))
        result = {}
        active = list(envs)
        while active:
            stats = self.sim.cloth_stats(active)
            nxt, targets = [], []
            for k, e in enumerate(active):
                if stats[k, 0] > 0.02:  # heights.min() > 0.02
                    result[e] = height[e]
                    continue
                height[e] += increment_step
                nxt.append(e)
                targets.append([[dist[e] / 2, height[e], -0.3], [-dist[e] / 2, height[e], -0.3]])
            self.movep(nxt, targets, speed=1e-3)
            active = []
            for e in nxt:
                if height[e] >= max_height:
                    result[e] = height[e]
                else:
                    active.append(e)
        return [result[e] for e in envs]

    # ---- simEnv.py:262-281
    def fling_primitive(self, envs, dist, fling_height, fling_speed):
        envs = [int(e) for e in envs]
        gh2 = self.grasp_height * 2

        def tg(y, z):
            return [[[d / 2, h if y is None else y, z], [-d / 2, h if y is None else y, z]] for d, h in zip(dist, fling_height)]

        self.movep(envs, tg(None, -0.2), speed=fling_speed)
        self.movep(envs, tg(None, 0.2), speed=fling_speed)
        self.movep(envs, tg(None, 0.2), speed=1e-2, min_steps=4)
        self.movep(envs, tg(gh2, -0.2), speed=1e-2)      # lower
        self.movep(envs, tg(gh2, -0.25), speed=5e-3)
        self.set_grasp(envs, False)                        # release
        self.reset_end_effectors(envs)

    def reset_end_effectors(self, envs):
        envs = [int(e) for e in envs]
        self.movep(envs, [[[0.5, 0.5, -0.5], [-0.5, 0.5, -0.5]]] * len(envs), speed=5e-3)

    # ---- simEnv.py:464-475: SimEnv.preaction / postaction around an action
    def preaction(self, envs=None):
        envs = [int(e) for e in (self.envs if envs is None else envs)]
        if envs:
            self.sim.snapshot_positions(envs)

    def postaction(self, envs=None, max_steps=300, tolerance=1e-2):
        """reset_end_effectors, wait_until_stable, and the "didn't really move cloth -> end early" test.  Returns the
        per-episode terminate flags (also kept in self.terminate)."""
        envs = [int(e) for e in (self.envs if envs is None else envs)]
        if not envs:
            return []
        self.reset_end_effectors(envs)
        _, steps = self.sim.wait_until_stable(envs, max_steps=max_steps, tolerance=tolerance)
        self.sim_steps += int(np.sum(steps))
        return self.postaction_check(envs)

    def postaction_check(self, envs):
        """simEnv.py:470-475: if the action didn't really move the cloth then end early."""
        envs = [int(e) for e in envs]
        deltas_max = self.sim.max_displacement(envs) if envs else []
        for e, d in zip(envs, deltas_max):
            if d < 5e-2:
                self.terminate[e] = True
        return [self.terminate[e] for e in envs]

    def act_scheduled(self, actions, envs=None, settle=True, cap_min=8, cap=64):
        """The action handlers AND postaction of many episodes, each running the reference's straight-line code as its own
        program (flingbot_amd/schedule.py) instead of the lock-step phases above: actions = {episode: (primitive, p1, p2,
        p1_grasp_cloth, p2_grasp_cloth)}; episodes of `envs` (default: all) without an entry only settle.  Same
        per-episode trajectories, bit for bit; returns ({episode: handler result}, terminate flags of envs)."""
        from . import schedule as sch

        envs = [int(e) for e in (self.envs if envs is None else envs)]
        progs = {}
        for e in envs:
            ep = sch.Episode(self, e)
            body = sch.PROGRAMS[actions[e][0]](ep, *actions[e][1:]) if e in actions else None
            progs[e] = sch.action_then_settle(ep, body) if settle else body
        out = sch.run_programs(self, {e: g for e, g in progs.items() if g is not None}, cap_min=cap_min, cap=cap)
        return out, (self.postaction_check(envs) if settle else [self.terminate[e] for e in envs])

    # ---- simEnv.py:283-318
    def pick_and_fling(self, p1, p2, p1_grasp_cloth, p2_grasp_cloth):
        """p1, p2: [n,3] grasp positions per episode (y is overwritten by grasp_height), *_grasp_cloth: bool[n].
        Returns a list of dicts per episode: {'dist', 'fling_height', 'terminated', 'skipped'}."""
        envs = [int(e) for e in self.envs]
        p1 = np.array(p1, np.float64).reshape(len(envs), 3)
        p2 = np.array(p2, np.float64).reshape(len(envs), 3)
        out = {e: dict(dist=None, fling_height=None, terminated=False, skipped=False) for e in envs}
        run = []
        for k, e in enumerate(envs):
            if not (p1_grasp_cloth[k] or p2_grasp_cloth[k]):
                out[e]["skipped"] = True  # both points not on cloth
            else:
                run.append(k)
        if run:
            idx = {envs[k]: k for k in run}
            act = [envs[k] for k in run]
            p1[:, 1] = self.grasp_height
            p2[:, 1] = self.grasp_height
            dist = {e: np.linalg.norm(np.array(p1[idx[e]]) - np.array(p2[idx[e]])) for e in act}
            self.movep(act, [[p1[idx[e]], p2[idx[e]]] for e in act])
            for e in act:  # only grasp points on cloth
                self.grasp_states[e] = [bool(p1_grasp_cloth[idx[e]]), bool(p2_grasp_cloth[idx[e]])]
            # lift to prefling
            self.movep(act, [[[dist[e] / 2, 0.3, -0.3], [-dist[e] / 2, 0.3, -0.3]] for e in act], speed=5e-3)
            grasped = self.is_cloth_grasped(act)
            keep = []
            for e, g in zip(act, grasped):
                if not g:
                    self.terminate[e] = True
                    out[e]["terminated"] = True
                else:
                    keep.append(e)
            if keep:
                d = self.stretch_cloth(keep, [dist[e] for e in keep], fling_height=0.3)
                if self.fixed_fling_height == -1:
                    h = self.lift_cloth(keep, d, fling_height=0.3)
                else:
                    h = [self.fixed_fling_height] * len(keep)
                self.fling_primitive(keep, d, h, self.fling_speed)
                for e, dd, hh in zip(keep, d, h):
                    out[e]["dist"], out[e]["fling_height"] = dd, hh
        return [out[e] for e in envs]

    # ---- simEnv.py:320-345 / :347-374: one-handed drag and pick-and-place; the second picker waits at (-0.2, 0.3, -0.2)
    _PARK = [-0.2, 0.3, -0.2]

    def _skip_unless(self, flags):
        envs = [int(e) for e in self.envs]
        out = {e: dict(skipped=not bool(flags[k])) for k, e in enumerate(envs)}
        return envs, out, [e for k, e in enumerate(envs) if flags[k]]

    def pick_and_drag(self, p1, p2, p1_grasp_cloth, p2_grasp_cloth=None):
        """SimEnv.pick_and_drag_primitive (simEnv.py:320-345) for every episode: p1 start, p2 end of the drag."""
        envs, out, act = self._skip_unless(p1_grasp_cloth)  # first grasp point not on cloth -> nothing happens
        if act:
            p1 = np.array(p1, np.float64).reshape(len(envs), 3)
            p2 = np.array(p2, np.float64).reshape(len(envs), 3)
            p1[:, 1] = self.grasp_height
            p2[:, 1] = self.grasp_height
            idx = {e: envs.index(e) for e in act}

            def at(pos, h=None):
                q = pos.copy()
                if h is not None:
                    q[1] = h
                return [q, self._PARK]

            self.movep(act, [at(p1[idx[e]], 0.3) for e in act], speed=5e-3)   # prestart
            self.movep(act, [at(p1[idx[e]]) for e in act], speed=5e-3)
            self.set_grasp(act, True)
            self.movep(act, [at(p2[idx[e]]) for e in act], speed=5e-3)
            self.set_grasp(act, False)
            self.movep(act, [at(p2[idx[e]], 0.3) for e in act], speed=5e-3)   # postend
            self.reset_end_effectors(act)
        return [out[e] for e in envs]

    def pick_and_place(self, p1, p2, p1_grasp_cloth, p2_grasp_cloth=None, lift_height=0.2):
        """SimEnv.pick_and_place_primitive (simEnv.py:347-374) for every episode."""
        envs, out, act = self._skip_unless(p1_grasp_cloth)
        if act:
            p1 = np.array(p1, np.float64).reshape(len(envs), 3)
            p2 = np.array(p2, np.float64).reshape(len(envs), 3)
            p1[:, 1] = self.grasp_height
            p2[:, 1] = self.grasp_height
            idx = {e: envs.index(e) for e in act}

            def at(pos, h=None):
                q = pos.copy()
                if h is not None:
                    q[1] = h
                return [q, self._PARK]

            self.movep(act, [at(p1[idx[e]], lift_height) for e in act], speed=5e-3)   # prepick
            self.movep(act, [at(p1[idx[e]]) for e in act], speed=5e-3)
            self.set_grasp(act, True)
            self.movep(act, [at(p1[idx[e]], lift_height) for e in act], speed=5e-3)
            self.movep(act, [at(p2[idx[e]], lift_height) for e in act], speed=5e-3)   # preplace
            self.movep(act, [at(p2[idx[e]]) for e in act], speed=5e-3)
            self.set_grasp(act, False)
            self.movep(act, [at(p2[idx[e]], lift_height) for e in act], speed=5e-3)
            self.reset_end_effectors(act)
        return [out[e] for e in envs]

    def pick_stretch_drag(self, p1, p2, p1_grasp_cloth, p2_grasp_cloth):
        """SimEnv.pick_stretch_drag_primitive (simEnv.py:376-428) for every episode."""
        envs, out, act = self._skip_unless([a or b for a, b in zip(p1_grasp_cloth, p2_grasp_cloth)])
        for o in out.values():
            o["dist"] = None
        if act:
            p1 = np.array(p1, np.float64).reshape(len(envs), 3)
            p2 = np.array(p2, np.float64).reshape(len(envs), 3)
            p1[:, 1] = self.grasp_height
            p2[:, 1] = self.grasp_height
            idx = {e: envs.index(e) for e in act}

            def raised(pos, h):
                q = pos.copy()
                q[1] = h
                return q

            self.movep(act, [[raised(p1[idx[e]], 0.3), raised(p2[idx[e]], 0.3)] for e in act])
            self.movep(act, [[p1[idx[e]], p2[idx[e]]] for e in act], speed=2e-3)
            for e in act:  # only grasp points on cloth
                self.grasp_states[e] = [bool(p1_grasp_cloth[idx[e]]), bool(p2_grasp_cloth[idx[e]])]
            dist = {e: np.linalg.norm(np.array(p1[idx[e]]) - np.array(p2[idx[e]])) for e in act}
            both = [e for e in act if all(self.grasp_states[e])]  # stretch if cloth is grasped by both
            if both:
                for e, d in zip(both, self.stretch_cloth(both, [dist[e] for e in both], fling_height=self.grasp_height)):
                    dist[e] = d
            targets_end, targets_post = [], []
            for e in act:
                drag_direction = np.cross(p1[idx[e]] - p2[idx[e]], np.array([0, 1, 0]))
                drag_direction = self.stretchdrag_dist * drag_direction / np.linalg.norm(drag_direction)
                left_start, right_start = self.picker_positions(e)  # float32 rows; + float64 direction -> float64
                left_end = left_start + drag_direction
                right_end = right_start + drag_direction
                left_end[1] += 0.1  # prevent ee go under cloth
                right_end[1] += 0.1
                left_post, right_post = left_end.copy(), right_end.copy()
                left_post[1] = 0.3
                right_post[1] = 0.3
                targets_end.append([left_end, right_end])
                targets_post.append([left_post, right_post])
                out[e]["dist"] = dist[e]
            self.movep(act, targets_end, speed=2e-3)
            self.set_grasp(act, False)
            self.movep(act, targets_post)
            self.reset_end_effectors(act)
        return [out[e] for e in envs]
