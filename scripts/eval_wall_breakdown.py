"""Development helper: where the wall time of evaluate.run_tasks goes on the host side -- inside fs_advance (the GPU runs,
the host waits), inside each host-side service (observe / act / coverage / snapshot / max_disp / stats / probe) and inside
the programs themselves (generator code between requests: fs_set_scene of a fresh task, state uploads, action selection).
usage: eval_wall_breakdown.py [tasks] [slots] [actions]"""
import collections, os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks, schedule as sch, evaluate
from flingbot_amd.env import BatchedFlingEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
PIPE = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # 0: blocking scheduler
CAP_MIN = int(sys.argv[5]) if len(sys.argv) > 5 else 0       # 0: run_tasks' default
CAP = int(sys.argv[6]) if len(sys.argv) > 6 else 0
random.seed(1); np.random.seed(1); torch.manual_seed(1)
tasks = []
for k in range(0, N, S):
    gen = fsim.FlingSim(n_envs=min(S, N - k), solver=0)
    tasks += ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(min(S, N - k))])
    gen.close()
tasks = [t for t in tasks if t is not None]        # (a rejected task is not part of the set, like the reference's generation loop)
ctx = fsim.FlingSim(n_envs=S, solver=0)
env = BatchedFlingEnv(ctx, episode_length=steps)
policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                 obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                 depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                 value_expl_decay=1.0, device="cuda:0")
acc, cnt = collections.Counter(), collections.Counter()

def timed(name, fn):
    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t; cnt[name] += 1
    return wrap

for name in ("advance", "advance_begin", "advance_end", "coverage", "snapshot_positions", "max_displacement", "observe_batch", "set_scene", "set_positions",
             "set_velocities", "cloth_stats", "stretch_probe"):
    if hasattr(ctx, name):
        setattr(ctx, name, timed("sim." + name, getattr(ctx, name)))
nets.prepare_image = timed("prepare_image", nets.prepare_image)
policy.act = timed("policy.act", policy.act)
env.selector.select = timed("selector.select", env.selector.select)
for fn in ("load_task_scene", "load_task_state"):
    import flingbot_amd.env as fenv
    setattr(fenv, fn, timed(fn, getattr(fenv, fn)))
t0 = time.perf_counter()
stats = evaluate.run_tasks(policy, env, tasks, pipeline=bool(PIPE), prebuild=bool(PIPE), cap_min=CAP_MIN or None, cap=CAP or None)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
flings = sum(stats["action_primitive_counts"].values())
print("pipeline %d cap_min %d cap %d: " % (PIPE, CAP_MIN, CAP), end="")
print("%d tasks / %d slots: %.2f s  %d flings (%.1f /s)  %d episode-steps (%.0f /s)" % (N, S, dt, flings, flings / dt, stats["simulation_steps"], stats["simulation_steps"] / dt))
rest = dt
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-26s %8.3f s  %5.1f %%  %6d calls  %.3f ms/call" % (k, v, 100 * v / dt, cnt[k], 1e3 * v / max(cnt[k], 1)))
nested = ("sim.set_scene", "sim.set_positions", "sim.set_velocities")  # inside load_task_*
top = sum(v for k, v in acc.items() if k not in nested)
print("  %-26s %8.3f s  %5.1f %%" % ("(everything else: python)", dt - top, 100 * (dt - top) / dt))
st = env.prim.sched_stats
print("  fs_advance calls %d, launch sequences %d, mean active %.1f" % (st["calls"], st["sequences"], st["episode_steps"] / max(st["sequences"], 1)))
at = ctx.advance_timing()
print("  inside fs_advance: wall %.2f s, device busy (first to last launch) %.2f s, before the first launch %.2f s, after the last launch (drain + results) %.2f s" % (
    at["wall_ms"] / 1e3, at["gpu_ms"] / 1e3, at["prep_ms"] / 1e3, (at["wall_ms"] - at["gpu_ms"] - at["prep_ms"]) / 1e3))
if not PIPE:
    print("  GPU idle while the loop ran: %.1f %% of the wall time" % (100 * (1 - at["gpu_ms"] / 1e3 / dt)))
