"""Development helper: how much of the evaluation loop's simulation time is lost to episodes waiting for each other.
Wraps FlingSim.movep / wait_until_stable: per call, launch sequences = max(iterations) while the useful work is
sum(iterations); prints totals and wall time per call type."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_episodes

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
random.seed(0); np.random.seed(0); torch.manual_seed(0)
gen = fsim.FlingSim(n_envs=E, solver=0)
tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(E)])
gen.close()
ctx = fsim.FlingSim(n_envs=E, solver=0)
acc = {"movep": [0, 0, 0, 0.0, 0], "wait": [0, 0, 0, 0.0, 0]}  # calls, launch sequences, episode-steps, seconds, slots
_movep, _wait = ctx.movep, ctx.wait_until_stable
def movep(envs, *a, **k):
    t0 = time.perf_counter(); it = _movep(envs, *a, **k); ctx.sync(); dt = time.perf_counter() - t0
    it = np.atleast_1d(it); r = acc["movep"]; r[0] += 1; r[1] += int(it.max()); r[2] += int(it.sum()); r[3] += dt; r[4] += int(it.max()) * len(it)
    return it
def wait(envs, *a, **k):
    t0 = time.perf_counter(); st, steps = _wait(envs, *a, **k); ctx.sync(); dt = time.perf_counter() - t0
    s = np.atleast_1d(steps); r = acc["wait"]; r[0] += 1; r[1] += int(s.max()); r[2] += int(s.sum()); r[3] += dt; r[4] += int(s.max()) * len(s)
    return st, steps
ctx.movep, ctx.wait_until_stable = movep, wait
env = BatchedFlingEnv(ctx, episode_length=3)
policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                 obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                 depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                 value_expl_decay=1.0, device="cuda:0")
t0 = time.perf_counter()
stats = run_episodes(policy, env, tasks)
dt = time.perf_counter() - t0
print("loop %.2f s, %d episode-steps" % (dt, stats["simulation_steps"]))
for k, (calls, seqs, work, sec, slots) in acc.items():
    print("%-6s calls %5d  launch sequences %6d  episode-steps %7d  batch efficiency %.2f  mean active %.1f  %.2f s (%.2f ms per sequence)" % (
        k, calls, seqs, work, work / max(slots, 1), work / max(seqs, 1), sec, sec / max(seqs, 1) * 1e3))
print("other (perception, selection, host): %.2f s" % (dt - sum(v[3] for v in acc.values())))
