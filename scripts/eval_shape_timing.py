"""Development helper: where the evaluation loop's time per frame comes from, WITHOUT the loop -- plain fs_step on launch lists
shaped like the loop's (DESIGN.md 6 "the 192-slot evaluation loop against the uniform benchmark"):
    A  the first E tasks of a stored set (scripts/make_task_set.py) in their stored, crumpled states: mixed sizes + contacts
    B  the same cloths laid out flat just above the ground: mixed sizes, no particle contacts
    C  E cloths of ONE size with about the same particle total, flat: no contacts, no size mix
    D  scripts/large_cloth_timing.py's figure: 64 flat 104 x 104 cloths popping out of the ground
ms per frame, particles per launch, G particle-iterations per second, and the candidate statistics of A.
usage: eval_shape_timing.py set.npz [E=132] [frames=20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import cloth_params
from flingbot_amd import sim as fsim, taskio, tasks as ftasks

path = sys.argv[1]
E = int(sys.argv[2]) if len(sys.argv) > 2 else 132
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
only = os.environ.get("SHAPE_ONLY", "ABCD")
tasks = taskio.TaskLoader(path, repeat=False).all_tasks()[:E]
E = len(tasks)
if os.environ.get("SHAPE_BALANCE"):
    # the list re-ordered so that the 8 XCD lanes (slot % 8, per launch chain of E / 2 slots) carry equal particle totals:
    # largest first, each to the lane with the smallest total so far that still has a free position
    order = sorted(range(E), key=lambda i: -int(tasks[i]["cloth_size"][0]) * int(tasks[i]["cloth_size"][1]))
    half = ((E + 15) // 16) * 8
    pos = [None] * E
    tot = {}
    free = {(c, x): [p for p in range(c * half, min((c + 1) * half, E)) if p % 8 == x] for c in range(2) for x in range(8)}
    for i in order:
        c, x = min((k for k in free if free[k]), key=lambda k: tot.get(k, 0))
        pos[free[(c, x)].pop(0)] = i
        tot[(c, x)] = tot.get((c, x), 0) + int(tasks[i]["cloth_size"][0]) * int(tasks[i]["cloth_size"][1])
    tasks = [tasks[i] for i in pos]
    print("list re-ordered for equal lane totals: lane totals %d .. %d particles" % (min(tot.values()), max(tot.values())), flush=True)
else:
    tot = {}
    half = ((E + 15) // 16) * 8
    for p, t in enumerate(tasks):
        k = (p // half, p % 8)
        tot[k] = tot.get(k, 0) + int(t["cloth_size"][0]) * int(t["cloth_size"][1])
    print("list in set order: lane totals %d .. %d particles" % (min(tot.values()), max(tot.values())), flush=True)


def timed(ctx, label, particles):
    ctx.step(3); ctx.sync()
    ms = []
    for _ in range(3):
        ctx.timer_start(); ctx.step(frames); ms.append(ctx.timer_stop() / frames)
    m = float(np.median(ms))
    print("%-72s %7.3f ms/frame  (%.3f .. %.3f)  %9d particles  %5.1f G particle-iterations/s  kernel form %d, %d chain(s)" % (
        label, m, min(ms), max(ms), particles, particles * 120 / m / 1e6, ctx.last_kernel_form(), ctx.last_stream_groups()), flush=True)
    return m


sizes = np.array([t["cloth_size"] for t in tasks])
total = int((sizes[:, 0] * sizes[:, 1]).sum())
if "A" in only or "B" in only:
    ctx = fsim.FlingSim(n_envs=E, solver=0)
    for e, t in enumerate(tasks):
        ftasks.load_task_scene(ctx, e, t)
    ctx.step(1)
    for e, t in enumerate(tasks):
        ftasks.load_task_state(ctx, e, t)
    if "A" in only:
        a = timed(ctx, "A  %d stored tasks, crumpled (sides %d..%d)" % (E, sizes.min(), sizes.max()), total)
        cnt = [np.asarray(ctx.get_last_neighbors(e)[0], np.int64) for e in range(0, E, 6)]
        allc = np.concatenate(cnt)
        per_wave = [np.concatenate([c, np.zeros((-c.size) % 64, np.int64)]).reshape(-1, 64) for c in cnt]
        wmax = np.concatenate([w.max(axis=1) for w in per_wave])
        print("   candidates: mean per particle %.2f, particles with any %.3f, a wave's longest list: mean %.2f (trips of four: %.2f), "
              "waves without any %.3f" % (allc.mean(), (allc > 0).mean(), wmax.mean(), np.ceil(wmax / 4.0).mean(), (wmax == 0).mean()), flush=True)
    if "B" in only:
        for e, t in enumerate(tasks):
            dx, dz = int(t["cloth_size"][0]), int(t["cloth_size"][1])
            p = np.array(ctx.get_positions(e), np.float32).reshape(-1, 4)
            gx, gz = np.meshgrid(np.arange(dx), np.arange(dz))
            p[:, 0] = (gx.ravel() - dx / 2) * 0.00625
            p[:, 2] = (gz.ravel() - dz / 2) * 0.00625
            p[:, 1] = 0.0125
            ctx.set_positions(e, p.ravel())
            ctx.set_velocities(e, np.zeros(3 * dx * dz, np.float32))
        timed(ctx, "B  the same cloths flat above the ground (no particle contacts)", total)
    ctx.close()
if "C" in only:
    side = int(round(np.sqrt(total / E)))
    ctx = fsim.FlingSim(n_envs=E, solver=0)
    for e in range(E):
        ctx.set_scene(e, cloth_params(side, side, pos=(0.0, -0.3, 0.0)))
    timed(ctx, "C  %d flat %d x %d cloths (one size, no contacts)" % (E, side, side), E * side * side)
    ctx.close()
if "D" in only:
    ctx = fsim.FlingSim(n_envs=64, solver=0)
    for e in range(64):
        ctx.set_scene(e, cloth_params(104, 104, pos=(0.0, -0.3, 0.0)))
    timed(ctx, "D  64 flat 104 x 104 cloths (scripts/large_cloth_timing.py)", 64 * 104 * 104)
    ctx.close()
