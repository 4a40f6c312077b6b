"""Development helper for PMC passes: 64 flat 64x64 sheets in free fall (no contacts, no ground) -- the iterate kernel's
springs and address arithmetic alone -- for 30 frames on the streaming back-end."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import cloth_params
from flingbot_amd import sim as fsim
E = 64
ctx = fsim.FlingSim(n_envs=E, solver=1)
p = cloth_params(64, 64, pos=(0.0, 1.0, 0.0))
for e in range(E):
    ctx.set_scene(e, p)
ctx.step(30); ctx.sync()
print("done", ctx.last_kernel_form())
