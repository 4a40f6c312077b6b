"""Development helper: host-side cost of putting a task into a slot (fs_set_scene + state upload + picker set-up), which
evaluate.run_tasks pays once per episode while the other slots wait for the scheduler."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from flingbot_amd import sim as fsim, tasks as ftasks
random.seed(0); np.random.seed(0)
n = 8
gen = fsim.FlingSim(n_envs=n, solver=0)
tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(n)])
gen.close()
ctx = fsim.FlingSim(n_envs=n, solver=0)
for rep in range(2):
    t_scene = t_state = 0.0
    for e, t in enumerate(tasks):
        t0 = time.perf_counter(); ftasks.load_task_scene(ctx, e, t); ctx.sync(); t1 = time.perf_counter()
        ctx.step_list([e], 1); ctx.sync(); t2 = time.perf_counter()
        ftasks.load_task_state(ctx, e, t); ctx.sync(); t3 = time.perf_counter()
        t_scene += t1 - t0; t_state += t3 - t2
        if rep == 1: print("task %d cloth %s: set_scene %.1f ms, state %.1f ms" % (e, list(t["cloth_size"]), (t1 - t0) * 1e3, (t3 - t2) * 1e3))
    print("rep %d: set_scene %.1f ms per task, state upload %.1f ms per task" % (rep, t_scene / n * 1e3, t_state / n * 1e3))
