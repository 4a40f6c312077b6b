import os, sys, time, random, cProfile, pstats
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
import numpy as np, torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_tasks
N, S = 96, 48
random.seed(0); np.random.seed(0); torch.manual_seed(0)
tasks = []
for k in range(0, N, S):
    gen = fsim.FlingSim(n_envs=S, solver=0)
    tasks += ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(S)])
    gen.close()
ctx = fsim.FlingSim(n_envs=S, solver=0)
env = BatchedFlingEnv(ctx, episode_length=3)
policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors), obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True, depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0, value_expl_decay=1.0, device="cuda:0")
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); stats = run_tasks(policy, env, tasks); pr.disable()
print("loop %.2f s" % (time.perf_counter() - t0))
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(22)
