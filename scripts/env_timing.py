"""Development helper: BatchedFlingEnv end to end -- E generated hard tasks (cloth sides 64..103), random-init fling policy,
a few env steps; wall time per stage group."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets, sim as fsim, tasks as ftasks
from flingbot_amd.env import BatchedFlingEnv

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_steps = 2
random.seed(0); np.random.seed(0); torch.manual_seed(0)
t0 = time.perf_counter()
gen = fsim.FlingSim(n_envs=E, solver=0)
tasks = ftasks.generate_tasks(gen, [ftasks.draw_task_parameters() for _ in range(E)])
gen.close()
t_gen = time.perf_counter() - t0
ctx = fsim.FlingSim(n_envs=E, solver=0)
env = BatchedFlingEnv(ctx, image_dim=int(sys.argv[2]) if len(sys.argv) > 2 else 400, episode_length=n_steps)
t0 = time.perf_counter(); obs = env.reset(tasks); torch.cuda.synchronize(); t_reset = time.perf_counter() - t0
net = nets.SpatialValueNet(rgb_only=True, device=env.device).to(env.device).eval().fold_batchnorm()
cov0 = np.array(ctx.coverage())
t_net = t_step = 0.0
acts = 0
for _ in range(n_steps):
    if not obs:
        break
    t0 = time.perf_counter()
    with torch.no_grad():
        vm = {e: {"fling": net(o).squeeze(1)} for e, o in obs.items()}
    torch.cuda.synchronize(); t1 = time.perf_counter()
    obs, rewards, term, actions = env.step(vm)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    t_net += t1 - t0; t_step += t2 - t1
    acts += sum(a is not None for a in actions.values())
cov1 = np.array(ctx.coverage())
print("E=%d: task generation %.1f s, reset %.1f s, value nets %.2f s, env.step x%d %.1f s (%d actions executed, %d simulation "
      "steps); coverage/flat area %.3f -> %.3f" % (E, t_gen, t_reset, t_net, n_steps, t_step, acts, env.prim.sim_steps,
      float(np.mean([cov0[e] / tasks[e]["flatten_area"] for e in env.envs])),
      float(np.mean([cov1[e] / tasks[e]["flatten_area"] for e in env.envs]))))
