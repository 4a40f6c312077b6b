"""Development helper: evaluate.run_tasks over a stored task set (scripts/make_task_set.py) -- the evaluation loop ALONE in
the process, so that rocprofv3's kernel trace / PMC passes see nothing of the task generation.
usage: eval_from_set.py tasks.npz [slots] [actions]
EVAL_CONTACTS=k   every k-th closed chunk, download the contact-candidate counts of 8 slots (perturbs the timing: own run)
EVAL_SOLVER=n     back-end of the context (0 AUTO)
EVAL_CAP_MIN / EVAL_CAP   chunk bounds of the pipelined scheduler (default 2 / 4)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, taskio
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_tasks

path = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 192
actions = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tasks = taskio.TaskLoader(path, repeat=False).all_tasks()
torch.manual_seed(1)
ctx = fsim.FlingSim(n_envs=S, solver=int(os.environ.get("EVAL_SOLVER", "0")))
env = BatchedFlingEnv(ctx, episode_length=actions)
policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=list(env.scale_factors),
                                 obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                 depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                 value_expl_decay=1.0, device="cuda:0")
every = int(os.environ.get("EVAL_CONTACTS", "0"))
samples = []          # (particles, mean candidates, max candidates, per-wave max summed / per-wave mean summed)
if every:
    inner, calls = ctx.advance_end, [0]

    def advance_end(*a, **k):
        r = inner(*a, **k)
        calls[0] += 1
        if calls[0] % every == 0:
            for e in range(calls[0] // every % 24, S, 24):
                n = ctx.n_particles(e)
                if n <= 0:
                    continue
                cnt, _ = ctx.get_last_neighbors(e)
                cnt = np.asarray(cnt[:n], dtype=np.int64)
                pad = (-n) % 64
                waves = np.concatenate([cnt, np.zeros(pad, np.int64)]).reshape(-1, 64)
                samples.append((n, cnt.mean(), cnt.max(), waves.max(axis=1).sum(), waves.mean(axis=1).sum(), (cnt > 0).mean()))
        return r
    ctx.advance_end = advance_end
t0 = time.perf_counter()
cap_min, cap = os.environ.get("EVAL_CAP_MIN"), os.environ.get("EVAL_CAP")
stats = run_tasks(policy, env, tasks, cap_min=int(cap_min) if cap_min else None, cap=int(cap) if cap else None)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
flings = sum(stats["action_primitive_counts"].values())
st = stats["scheduler"]
seq = max(st.get("sequences", 1), 1)
print("chunk bounds %s / %s: " % (cap_min or "default", cap or "default"), end="")
print("eval loop %d tasks / %d slots / %d actions: %.2f s  %d flings (%.1f /s)  %d episode-steps (%.0f /s)" % (
    len(tasks), S, actions, dt, flings, flings / dt, stats["simulation_steps"], stats["simulation_steps"] / dt), flush=True)
print("    fs_advance calls %d, launch sequences (frames) %d, mean active episodes %.1f, wall per frame %.3f ms" % (
    st.get("calls", 0), seq, st.get("episode_steps", 0) / seq, 1e3 * dt / seq), flush=True)
at = ctx.advance_timing()
print("    inside fs_advance: wall %.2f s, first-to-last launch %.2f s, before the first launch %.2f s" % (
    at["wall_ms"] / 1e3, at["gpu_ms"] / 1e3, at["prep_ms"] / 1e3), flush=True)
if samples:
    a = np.array(samples, dtype=np.float64)
    print("    contact candidates over %d sampled episode-frames: mean per particle %.2f, particles with any %.3f, mean of the "
          "episode maxima %.1f, largest %d; a wavefront's longest list / its mean list (what lock-step lanes pay) %.2f" % (
              len(a), (a[:, 1] * a[:, 0]).sum() / a[:, 0].sum(), (a[:, 5] * a[:, 0]).sum() / a[:, 0].sum(), a[:, 2].mean(),
              int(a[:, 2].max()), a[:, 3].sum() / max(a[:, 4].sum(), 1e-9)), flush=True)
ctx.close()
