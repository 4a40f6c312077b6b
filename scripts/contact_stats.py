"""Development helper: distribution of particle-contact candidate counts in the bench scenario, and what pass 2 of the fused
kernel makes of it: the contact set is ordered by count, thread t finishes entry t, so wave w pays for the longest list among
entries 64 w .. 64 w + 63."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from flingbot_amd import sim as fsim
E = 8
ctx = fsim.FlingSim(n_envs=E, solver=2)
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
for k in range(17):
    ctx.step(10)
    out = []
    for e in (0, 3, 7):
        cnt, _ = ctx.get_last_neighbors(e)
        out.append("%4d with contacts, mean %.2f max %d" % ((cnt > 0).sum(), cnt.mean(), cnt.max()))
    print("step %3d: " % (10 * (k + 1)) + " | ".join(out))
    if (k + 1) % 4 == 0:
        for e in (0, 3):
            cnt, _ = ctx.get_last_neighbors(e)
            s = np.sort(cnt[cnt > 0])[::-1]
            in_set = s[:1024]
            per_wave = [int(in_set[w * 64:(w + 1) * 64].max()) if in_set[w * 64:(w + 1) * 64].size else 0 for w in range(16)]
            print("   episode %d: set %d (inline %d), per-wave longest %s, sum %d, packed %.1f; histogram %s" % (
                e, in_set.size, s.size - in_set.size, per_wave, sum(per_wave), in_set.sum() / 64.0,
                np.bincount(np.minimum(s, 16))[1:].tolist()))
