"""Development helper: bit-exactness of the streaming back-end on cloths larger than the fused kernel takes (up to 104x104)
in a loose heap where neighbour lists reach the 96-entry cap; against the CPU oracle (slow: tens of seconds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import cloth_params
from flingbot_amd import sim as fsim
from oracle import OracleSim
# E identical episodes per launch: 1 -> the latency form of fs_k_iterate (uncompressed adjacency), 14 -> the throughput
# form with one-byte spring codes (launches above 32 x 4096 particles), 40 -> the grid form for the 104x104 case (launches of
# 96 x 4096 particles and more); episodes 0 and E-1 are compared with the oracle
E = int(sys.argv[1]) if len(sys.argv) > 1 else 14
for case, (dx, dz) in enumerate([(104, 104), (90, 70), (72, 100)]):
    ctx = fsim.FlingSim(n_envs=E, solver=0)
    hip, orc = ctx.env(0), OracleSim()
    for s in [ctx.env(e) for e in range(E)] + [orc]:
        s.set_scene(cloth_params(dx, dz, pos=(0.0, -0.3, 0.0)))
        r = np.random.RandomState(case)
        p = s.get_positions().reshape(-1, 4).copy()
        p[:, :3] = (r.rand(p.shape[0], 3) * [0.3, 0.12, 0.3] + [0, 0.03, 0]).astype(np.float32)   # loose heap: many contacts
        s.set_positions(p.ravel()); s.set_velocities(np.zeros(3 * p.shape[0], np.float32))
    ctx.step(6)
    orc.step(6)
    ok = np.array_equal(hip.get_positions().view(np.uint32), orc.get_positions().view(np.uint32))
    ok = ok and np.array_equal(ctx.env(E - 1).get_positions().view(np.uint32), orc.get_positions().view(np.uint32))
    ch, lh = ctx.get_last_neighbors(0); co, lo = orc.get_last_neighbors()
    mask = np.arange(96)[None, :] < co[:, None]
    ok = ok and np.array_equal(ch, co) and np.array_equal(np.where(mask, lh, -1), np.where(mask, lo, -1))
    print(dx, dz, "ok" if ok else "MISMATCH", "contacts max", co.max(), "mean %.1f" % co.mean(), flush=True)
