"""Development helper: the evaluation loop three ways on the same generated hard tasks (sides 64..103, reference sizes):
run_episodes with the lock-step primitives, run_episodes with per-episode programs inside each env.step, and run_tasks
(every slot on its own: no barrier between actions, slots refilled from the task queue).
usage: eval_async_timing.py [tasks] [slots] [actions]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_episodes, run_tasks

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
random.seed(0); np.random.seed(0); torch.manual_seed(0)
tasks = []
for k in range(0, N, S):  # generated in batches of S like the loop's context
    gen = fsim.FlingSim(n_envs=min(S, N - k), solver=0)
    tasks += ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(min(S, N - k))])
    gen.close()
ref = None
for label in ("lock-step", "scheduled steps", "async slots"):
    if label != "async slots" and (N > S or os.environ.get("EVAL_ONLY_ASYNC")):
        continue
    torch.manual_seed(0)
    ctx = fsim.FlingSim(n_envs=S, solver=int(os.environ.get("EVAL_SOLVER", "0")))  # 7: separate boundary kernels, 1: streaming only
    prims = os.environ.get("EVAL_PRIMITIVES", "fling").split(",")  # e.g. fling,stretchdrag,drag,place
    env = BatchedFlingEnv(ctx, action_primitives=prims, episode_length=steps, scheduled=label != "lock-step")
    policy = nets.MaximumValuePolicy(action_primitives=prims, num_rotations=12, scale_factors=list(env.scale_factors),
                                     obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                     depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                     value_expl_decay=1.0, device="cuda:0")
    t0 = time.perf_counter()
    stats = (run_tasks if label == "async slots" else run_episodes)(policy, env, tasks)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = ""
    if ref is None:
        ref = stats
    else:
        same = "  identical: %s" % (np.array_equal(ref["coverage_steps"], stats["coverage_steps"]) and ref["simulation_steps"] == stats["simulation_steps"])
    flings = sum(stats["action_primitive_counts"].values())
    print("%-16s %d tasks / %d slots: %.2f s  %d flings (%.1f /s)  %d episode-steps (%.0f /s)%s" % (
        label, N, S, dt, flings, flings / dt, stats["simulation_steps"], stats["simulation_steps"] / dt, same), flush=True)
    st = getattr(env.prim, "sched_stats", None)
    if st:
        print("    fs_advance calls %d, launch sequences %d, mean active %.1f" % (st["calls"], st["sequences"], st["episode_steps"] / max(st["sequences"], 1)), flush=True)
    ctx.close()
