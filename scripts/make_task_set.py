"""Development helper: generate `tasks` hard tasks (cloth sides 64..103, environment/tasks.py:105-275) in batches of `slots`
with seed `seed` -- the set bench.py's eval_loop leg draws -- and store them with flingbot_amd.taskio.save_tasks, so that
a profiler can trace the evaluation loop alone (python -m flingbot_amd.evaluate --tasks ..., scripts/eval_from_set.py).
usage: make_task_set.py [tasks] [slots] [seed] [out.npz]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import sim as fsim, tasks as ftasks, taskio

N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
S = int(sys.argv[2]) if len(sys.argv) > 2 else 192
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
out = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/tasks_%d.npz" % N
random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
t0 = time.perf_counter()
tasks = []
for k in range(0, N, S):
    part = [ftasks.draw_task_parameters() for _ in range(min(S, N - k))]
    gen = fsim.FlingSim(n_envs=len(part), solver=0)
    tasks += [t for t in ftasks.generate_tasks(gen, part) if t is not None]
    gen.close()
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
n = taskio.save_tasks(out, tasks)
sides = np.array([t["cloth_size"] for t in tasks])
print("%d tasks (sides %d..%d, mean particles %.0f) in %.1f s -> %s" % (
    n, sides.min(), sides.max(), float(np.mean(sides[:, 0] * sides[:, 1])), time.perf_counter() - t0, out), flush=True)
