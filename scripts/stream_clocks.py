"""Development helper: the last two launches of fs_k_iterate_gridl in a frame of the 64-episode streaming workload, workgroup by
workgroup (FS_BLOCK_CLOCKS build: `build_variant.sh clocks -DFS_BLOCK_CLOCKS`, FLINGSIM_LIB=variants/libfs_clocks.so): when each
workgroup entered and left, i.e. how long a workgroup runs, how long a launch lasts and what lies between two dependent launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from flingbot_amd import sim as fsim

E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = fsim.FlingSim(n_envs=E, solver=1)
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
ctx.step(80); ctx.sync()
for rep in range(2):
    ctx.step(1); ctx.sync()
    t = {94: [], 95: []}
    grp = {94: [], 95: []}
    longest, total = [], []
    for e in range(E):
        cnt, lists = ctx.get_last_neighbors(e)
        n = lists.shape[0]
        for row in (94, 95):
            v = lists[:, row].astype(np.int64) & 0xffffffff
            for b in range(0, n, 256):
                t[row].append((v[b], v[b + 1])); grp[row].append(e)
        for b in range(0, n, 256):
            longest.append(int(cnt[b:b + 256].max())); total.append(int(cnt[b:b + 256].sum()))
    longest, total = np.array(longest), np.array(total)
    print("frame %d, %d episodes, form %d, chains %d" % (81 + rep, E, ctx.last_kernel_form(), ctx.last_stream_groups()))
    for row in (94, 95):
        a = np.array(t[row], dtype=np.int64)
        dur = ((a[:, 1] - a[:, 0]) & 0xffffffff) / 100.0
        t0 = a[:, 0].min()
        print("   launch with flip %d: %d workgroups; first entry -> last exit %.2f us; entries spread over %.2f us; a workgroup runs %.2f us (median; min %.2f max %.2f)" % (
            row - 94, len(dur), ((a[:, 1] - t0) & 0xffffffff).max() / 100.0, ((a[:, 0] - t0) & 0xffffffff).max() / 100.0,
            np.median(dur), dur.min(), dur.max()))
    a = np.array(t[95], dtype=np.int64)
    dur = ((a[:, 1] - a[:, 0]) & 0xffffffff) / 100.0
    print("   workgroup duration vs the longest candidate list among its 256 particles: correlation %.2f (with the sum of its lists %.2f); by longest list: %s" % (
        np.corrcoef(dur, longest)[0, 1], np.corrcoef(dur, total)[0, 1],
        ", ".join("%d: %.2f us x%d" % (k, dur[longest == k].mean(), (longest == k).sum()) for k in sorted(set(longest.tolist())))))
    a94, a95 = np.array(t[94], dtype=np.int64), np.array(t[95], dtype=np.int64)
    first, second = (a94, a95) if a94[:, 0].min() < a95[:, 0].min() else (a95, a94)
    print("   between the two launches: last exit of the earlier -> first entry of the later %.2f us; first entry -> first entry %.2f us" % (
        ((second[:, 0].min() - first[:, 1].max()) & 0xffffffff) / 100.0 if second[:, 0].min() >= first[:, 1].max() else -((first[:, 1].max() - second[:, 0].min()) / 100.0),
        (second[:, 0].min() - first[:, 0].min()) / 100.0))
