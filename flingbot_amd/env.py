"""BatchedFlingEnv -- SimEnv.reset / step (environment/simEnv.py:477-515, 663-697) for many episodes on one GPU.

Composition of the device-side pieces of this package, one call per stage for ALL episodes where the stage allows it:
    tasks.load_tasks + FlingPrimitives.setup_pickers          SimEnv.reset           (simEnv.py:663-697)
    fs_render -> preprocess_obs -> fs_prepare_image           get_obs / prepare_image (simEnv.py:709-737, nets.py:177-193)
    ActionSelector.select (fs_select_action)                  get_max_value_valid_action (simEnv.py:560-661)
    FlingPrimitives.pick_and_fling / drag / place / stretchdrag   the action handlers  (simEnv.py:283-428)
    preaction / postaction, coverage                          SimEnv.step            (simEnv.py:464-515)
The observation stage (get_image's cv2.resize of the 720 x 720 render, the HSV cloth mask, its largest connected
component and the adaptive scale SimEnv.get_obs derives from it, simEnv.py:699-737) runs on the device in fs_observe
(csrc/fs_observe.hip); it follows OpenCV's / skimage's documented algorithms (oracle/observe.py) -- cv2 and skimage are
absent from this image, so that boundary is NOT pinned to the reference's own build.
The grasp-on-cloth flags (simEnv.py:235-255) test `depth != 2.0` on a Euclidean disc of conservative_grasp_radius, which
is pixel for pixel the filled cv2.circle (OpenCV's midpoint fill) for radii up to 6; the reference's default is 1.
Every other stage is pinned on its own in tests/ (see DESIGN.md section 2 and 4.3-4.8).
"""
import numpy as np
import torch

from . import nets
from .action import ActionSelector
from .primitives import FlingPrimitives
from .tasks import load_task_scene, load_task_state, load_tasks


class BatchedFlingEnv:
    def __init__(self, sim, action_primitives=("fling",), obs_dim=64, image_dim=400, num_rotations=12,
                 scale_factors=(1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75), pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5,
                 reach_distance_limit=1.2, conservative_grasp_radius=1, episode_length=10, grasp_height=0.02,
                 fling_speed=6e-3, stretchdrag_dist=0.3, device="cuda:0", render_dim=720, use_adaptive_scaling=True,
                 scheduled=True):
        self.sim = sim
        # scheduled: every episode runs its action + postaction as its own program on shared launch sequences
        # (flingbot_amd/schedule.py); False: the lock-step phases of FlingPrimitives.  Identical results.
        self.scheduled = bool(scheduled)
        self.actions = list(action_primitives)
        self.obs_dim, self.image_dim = int(obs_dim), int(image_dim)
        self.render_dim = int(render_dim)  # pyflex renders 720 x 720 (get_image reshapes to it, flex_utils.py:421)
        self.use_adaptive_scaling = bool(use_adaptive_scaling)
        if "fling" in self.actions:  # nets.py:213-218
            self.rotations = [(2 * i / (num_rotations - 1) - 1) * 90 for i in range(num_rotations)]
        else:
            self.rotations = [(2 * i / num_rotations - 1) * 180 for i in range(num_rotations)]
        self.scale_factors = np.array(scale_factors, np.float64)
        self.transformations = [(r, s) for r in self.rotations for s in self.scale_factors]  # product(rotations, scales)
        self.conservative_grasp_radius = int(conservative_grasp_radius)
        self.episode_length = int(episode_length)
        self.device = torch.device(device)
        self.selector = ActionSelector(self.actions, self.rotations, obs_dim, pix_grasp_dist, pix_drag_dist, pix_place_dist,
                                       reach_distance_limit, stretchdrag_dist=stretchdrag_dist, grasp_height=grasp_height)
        self._prim_kwargs = dict(grasp_height=grasp_height, fling_speed=fling_speed, stretchdrag_dist=stretchdrag_dist)
        self.envs, self.prim = [], None
        self.timestep, self.terminate = {}, {}
        self.pretransform_depth = {}
        self.pretransform_depth_dev = {}   # the same planes where fs_observe left them (views of the observation tensors)
        self.adaptive_scale_factors = {}

    # ---- SimEnv.reset for a batch of tasks (entry e of `tasks` becomes episode e)
    def reset(self, tasks):
        self.attach(load_tasks(self.sim, tasks))
        for e in self.envs:
            cp = self.sim.get_camera_params(e)
            self.sim.set_camera_params(e, [*cp[2:8], self.render_dim, self.render_dim])
        return self.observe()

    def attach(self, envs):
        """What SimEnv.reset does once the scene and state of the episodes are in place (simEnv.py:674-681): initial
        coverage, the two pickers at [0.2, 0.5, 0.0], reset_end_effectors, one simulation step, grasp off, counters."""
        self.envs = [int(e) for e in envs]
        self.init_coverage = np.array(self.sim.coverage())
        self.prim = FlingPrimitives(self.sim, self.envs, **self._prim_kwargs)
        self.prim.setup_pickers()
        self.timestep = {e: 0 for e in self.envs}
        self.terminate = {e: False for e in self.envs}

    def _adaptive_factors(self, bbox):
        """simEnv.py:722-731: scale factors shrunk to the cloth's bounding box (with some breathing room)."""
        factors = self.scale_factors.copy()
        if self.use_adaptive_scaling and bbox[4] > 0:
            dim = self.image_dim  # dimx == dimy
            cropx = max(dim - 2 * int(bbox[0]), dim - 2 * (dim - int(bbox[1])))
            cropy = max(dim - 2 * int(bbox[2]), dim - 2 * (dim - int(bbox[3])))
            crop = int(max(cropx, cropy) * 1.5)  # some breathing room
            if crop < dim:
                factors *= crop / dim
        return factors

    def get_obs(self, e):
        """SimEnv.get_obs (simEnv.py:710-737): render, resize to image_dim, cloth mask -> adaptive scale factors,
        preprocess_obs -- all in fs_observe; returns float32 [4, S, S] on the device."""
        obs, bbox = self.sim.observe(e, self.image_dim)
        self.pretransform_depth[e] = obs[3].cpu().numpy()
        self.pretransform_depth_dev[e] = obs[3]
        self.adaptive_scale_factors[e] = self._adaptive_factors(bbox)
        return obs

    def get_obs_batch(self, envs):
        """get_obs for several episodes: one fs_observe_batch call (the labelling rounds of all episodes share their host
        round trips) and ONE download of the depth planes the host side of the action selection reads."""
        envs = [int(e) for e in envs]
        obs, bbox = self.sim.observe_batch(envs, self.image_dim)
        depth = obs[:, 3].cpu().numpy() if envs else None
        for k, e in enumerate(envs):
            self.pretransform_depth[e] = depth[k]
            self.pretransform_depth_dev[e] = obs[k, 3].clone()  # (a view would keep the whole batch tensor alive per slot)
            self.adaptive_scale_factors[e] = self._adaptive_factors(bbox[k])
        return obs

    def get_transformations(self, e):
        return [(r, s) for r in self.rotations for s in self.adaptive_scale_factors[e]]  # product(rotations, scales)

    def observe(self):
        """{episode: transformed observation [T, 4, D, D] (CUDA)} for the episodes that are still running."""
        run = [e for e in self.envs if not self.terminate[e]]
        obs = self.get_obs_batch(run)
        return {e: nets.prepare_image(obs[k], self.get_transformations(e), self.obs_dim) for k, e in enumerate(run)}

    def _on_cloth(self, depth, pix):
        yy, xx = np.ogrid[:depth.shape[0], :depth.shape[1]]
        r = self.conservative_grasp_radius
        if r <= 0:
            return True
        disc = (yy - pix[0]) ** 2 + (xx - pix[1]) ** 2 <= r * r
        return bool((depth != 2.0)[disc].all())

    # ---- SimEnv.step for every running episode: value_maps[e] = {primitive: CUDA tensor [T, D, D]}
    def step(self, value_maps):
        run = [e for e in self.envs if not self.terminate[e] and e in value_maps]
        if not run:
            return {}, {}, dict(self.terminate), {}
        # the reference takes the snapshot and the coverage BEFORE selecting (simEnv.py:477-485); selection reads neither
        chosen = {}
        for e in run:
            action, params = self.selector.select(value_maps[e], self.adaptive_scale_factors[e], self.pretransform_depth[e],
                                                  depth_device=self.pretransform_depth_dev.get(e))
            if action is not None:
                d, pix = self.pretransform_depth[e], params["pretransform_pixels"]
                params["p1_grasp_cloth"] = self._on_cloth(d, (pix[0][1], pix[0][0]))
                params["p2_grasp_cloth"] = self._on_cloth(d, (pix[1][1], pix[1][0]))
                chosen[e] = (action, params)
        rewards, acted = self.step_actions(run, chosen)
        return self.observe(), rewards, dict(self.terminate), acted

    def step_actions(self, run, chosen):
        """SimEnv.step (simEnv.py:477-515) for the episodes `run` once the actions are known: chosen[e] = (primitive,
        {'p1', 'p2', 'p1_grasp_cloth', 'p2_grasp_cloth'}); an episode missing from `chosen` found no valid action.
        preaction -> coverage -> one batched primitive call per action type -> postaction (reset_end_effectors,
        wait_until_stable, "the cloth did not move" -> terminate) -> coverage -> timestep / episode_length.
        Returns ({episode: reward}, {episode: primitive or None})."""
        run = [int(e) for e in run]
        self.prim.preaction(run)
        prev = np.array(self.sim.coverage())
        if getattr(self, "scheduled", False):
            acts = {e: (chosen[e][0], chosen[e][1]["p1"], chosen[e][1]["p2"], chosen[e][1]["p1_grasp_cloth"],
                        chosen[e][1]["p2_grasp_cloth"]) for e in run if e in chosen}
            self.prim.act_scheduled(acts, run)
            return self._finish_step(run, chosen, prev)
        for action in self.actions:  # one batched primitive call per action type
            es = [e for e in run if e in chosen and chosen[e][0] == action]
            if not es:
                continue
            sub = FlingPrimitives(self.sim, es, **self._prim_kwargs)
            sub.grasp_states = {e: self.prim.grasp_states[e] for e in es}
            p1 = [chosen[e][1]["p1"] for e in es]
            p2 = [chosen[e][1]["p2"] for e in es]
            g1 = [chosen[e][1]["p1_grasp_cloth"] for e in es]
            g2 = [chosen[e][1]["p2_grasp_cloth"] for e in es]
            fn = {"fling": sub.pick_and_fling, "drag": sub.pick_and_drag, "place": sub.pick_and_place,
                  "stretchdrag": sub.pick_stretch_drag}[action]
            fn(p1, p2, g1, g2)
            self.prim.sim_steps += sub.sim_steps
            for e in es:
                self.prim.terminate[e] = self.prim.terminate[e] or sub.terminate[e]
                # SimEnv keeps grasp_states across the handler: an early return (cloth not grasped, simEnv.py:306-308) leaves
                # [p1_grasp, p2_grasp] set for postaction's reset_end_effectors / wait_until_stable
                self.prim.grasp_states[e] = list(sub.grasp_states[e])
        self.prim.postaction(run)
        return self._finish_step(run, chosen, prev)

    # ---- SimEnv.reset + SimEnv.step until the episode ends, for ONE slot, as a program (flingbot_amd/schedule.py)
    def open_slots(self, slots):
        """Bookkeeping for slots that episode_program will fill (instead of reset / attach)."""
        self.envs = [int(e) for e in slots]
        self.prim = FlingPrimitives(self.sim, self.envs, **self._prim_kwargs)
        self.timestep = {e: 0 for e in self.envs}
        self.terminate = {e: True for e in self.envs}
        self.init_coverage = np.zeros(self.sim.n_envs)
        self.unpaid_steps = 0  # simulation steps the lock-step path does not count either (the step inside set_scene)

    def episode_program(self, e, task, max_actions=None, prebuilt=None):
        """One episode in slot e -- SimEnv.reset (simEnv.py:663-697: set_scene(config, state), initial coverage, pickers,
        reset_end_effectors, one step, grasp off) and then SimEnv.step (simEnv.py:477-515) until it terminates -- written as
        the reference's straight-line code with a request wherever it needs the simulator, the policy or a reduction (see
        schedule.run_programs; evaluate.run_tasks provides the services "observe", "act", "coverage", "snapshot",
        "max_disp").  max_actions: the episode also ends after that many actions (None: episode_length alone).
        prebuilt: the task's sim.PrebuiltScene when its host half was built ahead (tasks.ScenePrebuilder).
        Returns {'coverage': [initial, after step 1, ...] (absolute areas), 'actions': [primitive or None, ...]}."""
        from . import schedule as sch

        e = int(e)
        sim, prim = self.sim, self.prim
        ep = sch.Episode(prim, e)
        load_task_scene(sim, e, task, prebuilt=prebuilt)
        yield ("step", 1)
        self.unpaid_steps += 1
        load_task_state(sim, e, task)
        cov = yield ("coverage",)
        self.init_coverage[e] = cov
        prim.grasp_states[e] = [False, False]
        prim.terminate[e] = False
        prim.place_pickers(e)
        yield from sch.reset_end_effectors(ep)
        yield ("step", 1)
        prim.set_grasp([e], False)
        cp = sim.get_camera_params(e)
        sim.set_camera_params(e, [*cp[2:8], self.render_dim, self.render_dim])
        self.timestep[e], self.terminate[e] = 0, False
        cov = yield ("coverage",)  # what run_sim's statistics call the initial coverage: the state the first observation shows
        rec = dict(coverage=[float(cov)], actions=[])
        obs = yield ("observe",)
        while True:
            maps = yield ("act", obs)
            action, params = self.selector.select(maps, self.adaptive_scale_factors[e], self.pretransform_depth[e],
                                                  depth_device=self.pretransform_depth_dev.get(e))
            body = None
            if action is not None:
                d, pix = self.pretransform_depth[e], params["pretransform_pixels"]
                g1 = self._on_cloth(d, (pix[0][1], pix[0][0]))
                g2 = self._on_cloth(d, (pix[1][1], pix[1][0]))
                body = sch.PROGRAMS[action](ep, params["p1"], params["p2"], g1, g2)
            yield ("snapshot",)                      # preaction
            prev = yield ("coverage",)
            yield from sch.action_then_settle(ep, body)
            moved = yield ("max_disp",)
            if moved < 5e-2:  # if didn't really move cloth then end early (simEnv.py:470-475)
                prim.terminate[e] = True
            curr = yield ("coverage",)
            self.timestep[e] += 1
            limit = self.episode_length if max_actions is None else min(self.episode_length, int(max_actions))
            self.terminate[e] = prim.terminate[e] or self.timestep[e] >= limit
            rec["coverage"].append(float(curr))
            rec["actions"].append(action)
            rec.setdefault("rewards", []).append(float(curr - prev))
            rec.setdefault("preaction_coverage", []).append(float(prev))
            if self.terminate[e]:
                return rec
            obs = yield ("observe",)

    def _finish_step(self, run, chosen, prev):
        curr = np.array(self.sim.coverage())
        rewards = {}
        for e in run:
            self.timestep[e] += 1
            self.terminate[e] = self.prim.terminate[e] or self.timestep[e] >= self.episode_length
            rewards[e] = float(curr[e] - prev[e])
        return rewards, {e: chosen.get(e, (None, None))[0] for e in run}
