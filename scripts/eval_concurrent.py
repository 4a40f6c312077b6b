"""Development helper: the evaluation loop with the episodes split over G independent contexts (own HIP stream each) driven
by G host threads -- small launches are bound by the ~8 us dependent-launch interval of ONE stream, so several streams
overlap where one cannot fill the chip.  Same tasks, same policy; reports wall time per G."""
import os, sys, time, random, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks
from flingbot_amd.env import BatchedFlingEnv
from flingbot_amd.evaluate import run_episodes

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
groups = [int(g) for g in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4]
random.seed(0); np.random.seed(0); torch.manual_seed(0)
gen = fsim.FlingSim(n_envs=E, solver=0)
tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(E)])
gen.close()
policy = nets.MaximumValuePolicy(action_primitives=["fling"], num_rotations=12, scale_factors=[1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75],
                                 obs_dim=64, pix_grasp_dist=8, pix_drag_dist=8, pix_place_dist=5, rgb_only=True,
                                 depth_only=False, action_expl_prob=0.0, action_expl_decay=1.0, value_expl_prob=0.0,
                                 value_expl_decay=1.0, device="cuda:0")
for net in policy.value_nets.values():
    net.fold_batchnorm()
ref = None
for G in groups:
    per = E // G
    ctxs = [fsim.FlingSim(n_envs=per, solver=0) for _ in range(G)]
    envs = [BatchedFlingEnv(c, episode_length=steps) for c in ctxs]
    out = [None] * G

    def work(g):
        out[g] = run_episodes(policy, envs[g], tasks[g * per:(g + 1) * per], fold=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    final = np.concatenate([o["final_coverage"] for o in out])
    sims = sum(o["simulation_steps"] for o in out)
    if ref is None:
        ref = final
    print("G=%d contexts x %d episodes: %.2f s, %d episode-steps (%.0f /s), final coverage %.4f, identical to G=%d: %s" % (
        G, per, dt, sims, sims / dt, final.mean(), groups[0], bool(np.array_equal(final, ref))), flush=True)
    for c in ctxs: c.close()
