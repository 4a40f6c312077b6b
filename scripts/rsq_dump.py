"""EXPERIMENTS R4.1 helper: positions of 4 bench episodes after 1 / 10 / 100 frames with the library FLINGSIM_LIB selects
-> gpurun_out/rsq/<tag>.npz (compare two tags with rsq_compare below)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from flingbot_amd import sim as fsim

tag = sys.argv[1]
if tag == "compare":
    a, b = (np.load(f"gpurun_out/rsq/{t}.npz") for t in sys.argv[2:4])
    for k in a.files:
        d = np.abs(a[k] - b[k])[:, :3]
        print(k, "max abs diff %.3e" % d.max(), "rel to extent %.3e" % (d.max() / np.abs(a[k][:, :3]).max()))
    sys.exit(0)
ctx = fsim.FlingSim(n_envs=4, solver=2)
for e in range(4):
    bench.setup_episode(ctx.env(e), seed=e)
out, done = {}, 0
for f in (1, 10, 100):
    ctx.step(f - done); done = f
    for e in range(4):
        out[f"f{f}_e{e}"] = ctx.get_positions(e).reshape(-1, 4)
os.makedirs("gpurun_out/rsq", exist_ok=True)
np.savez(f"gpurun_out/rsq/{tag}.npz", **out)
print("dumped", tag)
