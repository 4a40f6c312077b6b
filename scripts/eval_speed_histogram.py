"""Development helper for the costing of neighbour-list reuse (EXPERIMENTS.md): how fast do the cloths of the evaluation loop
move?  After every fs_advance chunk (blocking scheduler, 4-step chunks) the largest velocity component of every episode
that took part is sampled (fs_cloth_stats) and weighted with the steps it took; prints the share of episode-steps below a
few speeds and the number of substeps a candidate superset with skin s would survive (s / 2 of accumulated displacement at
h = 2.5 ms per substep).
usage: eval_speed_histogram.py [tasks] [slots] [actions]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks, evaluate
from flingbot_amd.env import BatchedFlingEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
random.seed(1); np.random.seed(1); torch.manual_seed(1)
tasks = []
for k in range(0, N, S):
    gen = fsim.FlingSim(n_envs=min(S, N - k), solver=0)
    tasks += ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(min(S, N - k))])
    gen.close()
ctx = fsim.FlingSim(n_envs=S, solver=0)
env = BatchedFlingEnv(ctx, episode_length=steps)
policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                 obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                 depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                 value_expl_decay=1.0, device="cuda:0")
speeds, weights = [], []
adv = ctx.advance
def advance(envs, *a, **k):
    prog, status, st = adv(envs, *a, **k)
    v = ctx.cloth_stats(list(envs))[:, 2]
    speeds.extend(v.tolist()); weights.extend(st.tolist())
    return prog, status, st
ctx.advance = advance
stats = evaluate.run_tasks(policy, env, tasks, pipeline=False, prebuild=False, cap_min=4, cap=4)
v, w = np.array(speeds), np.array(weights, float)
tot = w.sum()
print("%d episode-steps sampled (largest |velocity component| of the cloth after each 4-step chunk)" % tot)
for thr in (0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0):
    print("  max |v| < %.2f m/s: %5.1f %% of the episode-steps" % (thr, 100 * w[v < thr].sum() / tot))
r = 0.01125
for skin in (r / 2, r):
    life = np.maximum(1.0, (skin / 2) / (np.maximum(v, 1e-6) * 2.5e-3))  # substeps until skin / 2 is used up at this speed
    # per substep: a rebuild every `life` substeps at (1 + skin / r)^3 times the cost of a search, a filter pass (~1/4) otherwise
    cost = ((1 + skin / r) ** 3) / life + 0.25 * (1 - 1 / life)
    cost = np.minimum(cost, 1.0)  # never worse than searching every substep (the episode would keep doing that)
    print("  skin %.4f m: mean search cost %.2f of today's (1.0 = no gain)" % (skin, (cost * w).sum() / tot))
