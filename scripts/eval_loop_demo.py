"""Development helper: BASELINE.json configs[4] in miniature -- E generated hard tasks, a randomly initialised
MaximumValuePolicy (no flingbot.pth in this image), evaluate.run_episodes; wall time and the reference's statistics."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_episodes

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
image_dim = int(sys.argv[3]) if len(sys.argv) > 3 else 400   # the 720 x 720 render is resized to this (reference: image_dim)
random.seed(0); np.random.seed(0); torch.manual_seed(0)
gen = fsim.FlingSim(n_envs=E, solver=0)
tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(E)])
gen.close()
ctx = fsim.FlingSim(n_envs=E, solver=0)
env = BatchedFlingEnv(ctx, image_dim=image_dim, episode_length=steps)
policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                 obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                 depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                 value_expl_decay=1.0, device="cuda:0")
t0 = time.perf_counter()
stats = run_episodes(policy, env, tasks)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("E=%d episodes x %d actions: %.1f s, %d simulation steps (%.0f episode-steps/s), %s" % (
    E, steps, dt, stats["simulation_steps"], stats["simulation_steps"] / dt, stats["action_primitive_counts"]))
print("  " + "  ".join("%s %.3f" % kv for kv in stats["mean"].items()))
