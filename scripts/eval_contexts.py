"""Development helper: evaluate.run_tasks with the slots split over G contexts (own HIP stream(s) each), one host thread per
context, tasks dealt round-robin -- does one context's GPU work fill the other's host-side gaps?
usage: eval_contexts.py [tasks] [slots] [actions] [G,G,...] [chains per context, 0 = default]"""
import os, sys, time, random, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_tasks

N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
groups = [int(g) for g in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2]
chains = int(sys.argv[5]) if len(sys.argv) > 5 else 0
random.seed(1); np.random.seed(1); torch.manual_seed(1)
tasks = []
for k in range(0, N, S):
    gen = fsim.FlingSim(n_envs=min(S, N - k), solver=0)
    tasks += ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(min(S, N - k))])
    gen.close()
policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=[1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75],
                                 obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                 depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                 value_expl_decay=1.0, device="cuda:0")
for net in policy.value_nets.values():
    net.fold_batchnorm()
ref = None
for G in groups:
    per = S // G
    ctxs = [fsim.FlingSim(n_envs=per, solver=0) for _ in range(G)]
    if chains:
        for c in ctxs:
            c.set_stream_groups(chains)
    envs = [BatchedFlingEnv(c, episode_length=steps) for c in ctxs]
    out = [None] * G

    def work(g):
        out[g] = run_tasks(policy, envs[g], tasks[g::G], fold=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    final = np.empty(N)
    for g in range(G):
        final[g::G] = out[g]["final_coverage"]
    sims = sum(o["simulation_steps"] for o in out)
    flings = sum(sum(o["action_primitive_counts"].values()) for o in out)
    if ref is None:
        ref = final
    print("G=%d contexts x %d slots (chains %d): %.2f s, %d flings (%.1f /s), %d episode-steps (%.0f /s), identical to the first: %s" % (
        G, per, chains, dt, flings, flings / dt, sims, sims / dt, bool(np.array_equal(final, ref))), flush=True)
    for c in ctxs: c.close()
