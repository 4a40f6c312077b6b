"""Development helper: the evaluation loop (scripts/eval_loop_demo.py's set-up) with the lock-step primitives and with the
per-episode programs of flingbot_amd/schedule.py at several chunk bounds; same tasks, same policy -- the statistics must be
identical, the wall time is what differs.  usage: eval_schedule_timing.py [episodes] [actions]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, primitives, sim as fsim, tasks as ftasks
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_episodes

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
random.seed(0); np.random.seed(0); torch.manual_seed(0)
gen = fsim.FlingSim(n_envs=E, solver=0)
tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(E)])
gen.close()
ref = None
_act = primitives.FlingPrimitives.act_scheduled
for label, scheduled, caps in (("lock-step", False, None), ("scheduled 8..64", True, (8, 64)), ("scheduled 4..32", True, (4, 32)),
                               ("scheduled 16..64", True, (16, 64)), ("scheduled 2..16", True, (2, 16)), ("scheduled 8..128", True, (8, 128))):
    if caps:
        primitives.FlingPrimitives.act_scheduled = (lambda c: lambda self, a, envs=None, settle=True, **k: _act(self, a, envs, settle, cap_min=c[0], cap=c[1]))(caps)
    torch.manual_seed(0)
    ctx = fsim.FlingSim(n_envs=E, solver=0)
    env = BatchedFlingEnv(ctx, episode_length=steps, scheduled=scheduled)
    policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                     obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                     depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                     value_expl_decay=1.0, device="cuda:0")
    t0 = time.perf_counter()
    stats = run_episodes(policy, env, tasks)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = ""
    if ref is None:
        ref = stats
    else:
        same = "  identical to lock-step: %s" % (np.array_equal(ref["coverage_steps"], stats["coverage_steps"]) and
                                                 ref["simulation_steps"] == stats["simulation_steps"])
    print("%-18s %.2f s  %d episode-steps (%.0f /s)  %s%s" % (label, dt, stats["simulation_steps"], stats["simulation_steps"] / dt,
                                                             stats["action_primitive_counts"], same), flush=True)
    if scheduled:
        st = env.prim.sched_stats
        print("    fs_advance calls %d, launch sequences %d, mean active %.1f, slot efficiency %.2f" % (
            st["calls"], st["sequences"], st["episode_steps"] / max(st["sequences"], 1), st["episode_steps"] / max(st["slots"], 1)), flush=True)
    ctx.close()
