"""Development helper: the pipelined evaluation loop (chunks queued ahead, service lane, prebuilt scenes) against the
blocking scheduler and the lock-step loop on randomized task sets -- all four primitives in the action space, cloth sides
30..60, different slot counts, chunk bounds and pipeline depths.  Statistics must be identical.  Not part of the
test-suite (minutes).  usage: soak_pipeline.py [rounds]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks, schedule as sch
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_episodes, run_tasks

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bad = 0
for rnd in range(rounds):
    seed = 100 + rnd
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    n = 8 + 2 * rnd
    gen = fsim.FlingSim(n_envs=n, solver=0)
    tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters(min_cloth_size=30, strict_min_edge_length=30, max_cloth_size=60)
                                        for _ in range(n)])
    gen.close()
    actions = ("fling", "stretchdrag", "drag", "place") if rnd % 2 else ("fling",)
    ref = None
    for label, slots, kw in (("lock-step", n, None), ("blocking", 3 + rnd, dict(pipeline=False, prebuild=False)),
                             ("pipelined 2/4", 3 + rnd, dict()), ("pipelined 1/1", 2 + rnd, dict(cap_min=1, cap=1)),
                             ("pipelined 4/16", n, dict(cap_min=4, cap=16))):
        ctx = fsim.FlingSim(n_envs=slots, solver=0)
        env = BatchedFlingEnv(ctx, action_primitives=actions, image_dim=128, episode_length=3)
        torch.manual_seed(seed)
        policy = nets.MaximumValuePolicy(action_primitives=list(actions), num_rotations=12, scale_factors=list(env.scale_factors),
                                         obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                         depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                         value_expl_decay=1.0, device="cuda:0")
        t0 = time.perf_counter()
        stats = run_episodes(policy, env, tasks) if kw is None else run_tasks(policy, env, tasks, **kw)
        dt = time.perf_counter() - t0
        same = True
        if ref is None:
            ref = stats
        else:
            same = (np.array_equal(ref["coverage_steps"], stats["coverage_steps"]) and ref["simulation_steps"] == stats["simulation_steps"]
                    and ref["action_primitive_counts"] == stats["action_primitive_counts"] and np.array_equal(ref["episode_length"], stats["episode_length"]))
            bad += not same
        print("round %d  %-15s %2d tasks / %2d slots  %s  %.2f s  %d steps  %s" % (
            rnd, label, n, slots, "+".join(a[0] for a in actions), dt, stats["simulation_steps"], "identical" if same else "DIFFERENT"), flush=True)
        assert ctx.advance_in_flight() == 0
        ctx.close()
print("soak_pipeline: %d mismatches" % bad)
sys.exit(1 if bad else 0)
