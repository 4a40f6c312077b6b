"""Development helper: batched generation of E 'hard' tasks (64x64 ... 104x104 grid cloths) on the device."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from flingbot_amd import sim as fsim, tasks as ftasks
E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
random.seed(0); np.random.seed(0)
params = [ftasks.draw_task_parameters() for _ in range(E)]
ctx = fsim.FlingSim(n_envs=E, solver=0)
t0 = time.perf_counter()
out = ftasks.generate_hard_tasks(ctx, params)
dt = time.perf_counter() - t0
ok = [t for t in out if t is not None]
print("%d hard tasks (cloth sides 64..103) in %.1f s -> %.2f tasks/s; %d kept; mean coverage/flat area %.3f" % (
    E, dt, E / dt, len(ok), float(np.mean([t["initial_coverage"] / t["flatten_area"] for t in ok]))))
