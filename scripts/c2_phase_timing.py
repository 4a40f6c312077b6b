"""Development helper: where the C2 workload's time goes -- the scripted fling of bench.py leg by leg (ms per simulation step
of E episodes, HIP-event time), next to plain steps of the crumpled state it starts from.  usage: c2_phase_timing.py [E]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from flingbot_amd import sim as fsim
from flingbot_amd.primitives import FlingPrimitives

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = fsim.FlingSim(n_envs=E, solver=0)
bench.c2_crumple(ctx, list(range(E)))
ctx.sync()
ctx.timer_start(); ctx.step(20); ms = ctx.timer_stop()
print("crumpled by C2's recipe, no pickers: %.3f ms per step (one launch of 20 frames)" % (ms / 20))
ctx.timer_start()
for _ in range(20):
    ctx.step(1)
ms = ctx.timer_stop()
print("                                     %.3f ms per step (20 launches of one frame)" % (ms / 20))
prim = FlingPrimitives(ctx, range(E))
for e in range(E):
    prim.place_pickers(e)
ctx.timer_start(); ctx.step(20); ms = ctx.timer_stop()
print("the same with the two pickers parked above: %.3f ms per step (one launch of 20 frames)" % (ms / 20))
ctx.timer_start()
for _ in range(20):
    ctx.step(1)
ms = ctx.timer_stop()
print("                                            %.3f ms per step (20 launches of one frame)" % (ms / 20))
stay = np.stack([ctx.get_shape_states(e).reshape(-1, 14)[:, :3] for e in range(E)]).astype(np.float64)
stay[:, :, 1] += 1e-3 * 20
ctx.timer_start(); ctx.movep(np.arange(E, dtype=np.int32), stay, np.zeros((E, 2), int), speed=1e-3, limit=2000); ms = ctx.timer_stop()
print("                                            %.3f ms per step (movep of 20 steps, pickers creeping upwards, nothing grasped)" % (ms / max(ctx.last_movep_steps / E, 1)))
corners = np.stack([ctx.get_positions(e).reshape(-1, 4)[[0, 63], :3] for e in range(E)]).astype(np.float64)
envs = np.arange(E, dtype=np.int32)


def leg(name, targets, grasp, speed):
    ctx.sync(); t0 = time.perf_counter()
    ctx.timer_start()
    ctx.movep(envs, targets, np.full((E, 2), int(grasp)), speed=speed, limit=2000)
    ms = ctx.timer_stop(); wall = (time.perf_counter() - t0) * 1e3
    steps = ctx.last_movep_steps / E
    if steps:
        nl = [int((c > 0).sum()) for c in (ctx.get_last_neighbors(e)[0] for e in (0, E // 2))]
        print("%-28s %4.0f steps per episode  %.3f ms per step (GPU)  %.3f (wall)   particles with candidates: %s" % (name, steps, ms / steps, wall / steps, nl))


c = corners
above = c.copy(); above[:, :, 1] += 0.05
leg("to above the corners", above, False, 0.05)
on = c.copy(); on[:, :, 1] += 0.01
leg("down onto the corners", on, False, 5e-3)
up = on.copy(); up[:, :, 1] = 0.3
leg("lift to 0.3", up, True, 5e-3)
fwd = up.copy(); fwd[:, :, 2] += 0.2
leg("forward", fwd, True, 6e-3)
back = up.copy(); back[:, :, 2] -= 0.2
leg("back", back, True, 6e-3)
low = back.copy(); low[:, :, 1] = 0.05
leg("lower", low, True, 6e-3)
leg("release", low, False, 6e-3)
for k in range(3):
    ctx.timer_start(); ctx.step(100); ms = ctx.timer_stop()
    print("settle %d-%d: %.3f ms per step" % (100 * k, 100 * k + 100, ms / 100))
