"""Development helper: FlingSim.observe (fs_observe: render 720^2 -> resize -> cloth mask -> largest component -> bbox ->
observation tensor, all on the device) against the host path over pyflex.render's download (numpy restatement)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from flingbot_amd import sim as fsim
from oracle import observe as oo   # development comparison only

S = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ctx = fsim.FlingSim(n_envs=1, solver=0)
bench.setup_episode(ctx.env(0), 0)
ctx.step(60)
ctx.observe(0, S); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    obs, bbox = ctx.observe(0, S)
torch.cuda.synchronize(); dev = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(3):
    rgba, depth = ctx.render(0)
    ref = oo.get_obs(rgba, depth, 720, S)
host = (time.perf_counter() - t0) / 3
print("observe 720^2 -> %d^2: device %.2f ms per call (bbox %s), host numpy restatement over the downloaded frame %.1f ms" % (
    S, dev * 1e3, bbox.tolist(), host * 1e3))
