import os, sys, time, random, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np
from flingbot_amd import sim as fsim, tasks as ftasks
random.seed(0); np.random.seed(0)
S = 48
gen = fsim.FlingSim(n_envs=S, solver=0)
params = [ftasks.draw_task_parameters() for _ in range(S)]
pr = cProfile.Profile(); t0 = time.perf_counter()
pr.enable(); tasks = ftasks.generate_tasks(gen, params); pr.disable()
print("generation %.2f s for %d tasks" % (time.perf_counter() - t0, S))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
