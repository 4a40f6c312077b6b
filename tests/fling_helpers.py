"""Shared by the fling-primitive tests: the golden cases and a CPU stand-in for FlingSim built on the oracle (test
infrastructure only), so `flingbot_amd.primitives.FlingPrimitives` -- pure host logic -- can be checked without a GPU."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def picker_centres():
    """Picker.reset([0, 0.1, 0]) sphere centres (flex_utils.py:82-97)."""
    r = np.sqrt(2 - 1) * 0.02 * 2.
    return [[0.0 + np.cos(2 * np.pi * i / 2) * r, 0.1, 0.0 + np.sin(2 * np.pi * i / 2) * r] for i in range(2)]


def load_fling_golden():
    return np.load(os.path.join(GOLD, "fling_golden.npz"))


class OracleBatch:
    """The slice of the FlingSim interface FlingPrimitives uses, on N independent CPU oracles + the numpy restatement of
    the reference picker (oracle/picker.py).  The reductions are the reference's own numpy expressions."""

    def __init__(self, n, scene_params, init_pos):
        from oracle import OracleSim
        from oracle.picker import OraclePicker

        self.sims, self.tools = [], []
        for _ in range(n):
            o = OracleSim()
            o.set_scene(scene_params)
            o.step(1)
            o.set_positions(init_pos.ravel())
            o.set_velocities(np.zeros(3 * init_pos.shape[0], np.float32))
            t = OraclePicker(o)
            t.reset(picker_centres())
            self.sims.append(o)
            self.tools.append(t)

    def movep(self, envs, targets, grasp, speed=0.1, limit=1000, min_steps=None, eps=1e-4):
        targets = np.asarray(targets)
        return np.array([self.tools[e].movep(targets[k], [bool(g) for g in grasp[k]], speed=speed, limit=limit,
                                             min_steps=min_steps, eps=eps) for k, e in enumerate(envs)], np.int32)

    def get_shape_states(self, e):
        return self.sims[e].get_shape_states()

    def get_positions(self, e):
        return self.sims[e].get_positions()

    def cloth_stats(self, envs):
        out = np.empty((len(envs), 3), np.float32)
        for k, e in enumerate(envs):
            pos = self.sims[e].get_positions().reshape(-1, 4)
            out[k] = (pos[:, 1].min(), pos[:, 1].max(), np.abs(self.sims[e].get_velocities()).max())
        return out

    def stretch_probe(self, envs, midpoints_xz, height_thr):
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
        stable, steps = [], []
        for e in envs:
            done, ok = 0, False
            for _ in range(max_steps):  # flex_utils.py:430-441
                if np.abs(self.sims[e].get_velocities()).max() < tolerance:
                    ok = True
                    break
                self.sims[e].step(1)
                done += 1
            stable.append(ok)
            steps.append(done)
        return np.array(stable), np.array(steps, np.int32)


def load_primitives_golden():
    return np.load(os.path.join(GOLD, "primitives_golden.npz"))


def run_primitives_golden(make_sim, get_positions, get_shapes):
    """Runs the golden cases of the other manipulation primitives (drag / place / stretchdrag), batched per kind, on a
    simulator made by make_sim(n) and checks final particle positions and picker states bit for bit."""
    from flingbot_amd.primitives import FlingPrimitives

    g = load_primitives_golden()
    kinds = [str(k) for k in g["kind"]]
    for kind in ("drag", "place", "stretchdrag"):
        cases = [c for c, k in enumerate(kinds) if k == kind]
        sim = make_sim(len(cases))
        prim = FlingPrimitives(sim, range(len(cases)), stretchdrag_dist=float(g["stretchdrag_dist"]))
        fn = {"drag": prim.pick_and_drag, "place": prim.pick_and_place, "stretchdrag": prim.pick_stretch_drag}[kind]
        out = fn(g["p1"][cases], g["p2"][cases], g["g1"][cases], g["g2"][cases])
        for k, c in enumerate(cases):
            assert out[k]["skipped"] == (g["steps"][c] == 0), (kind, c)
            if kind == "stretchdrag" and not np.isnan(g["stretch_ret"][c]):
                assert out[k]["dist"] == g["stretch_ret"][c], (kind, c)
            assert np.array_equal(get_positions(sim, k).view(np.uint32), g["pos"][c].view(np.uint32)), (kind, c)
            assert np.array_equal(np.asarray(get_shapes(sim, k), np.float32).view(np.uint32), g["shapes"][c].view(np.uint32)), (kind, c)
