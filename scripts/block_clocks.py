"""Development helper: how long every workgroup (= episode) of one launch of the fused bench kernel runs.  Needs the
FS_BLOCK_CLOCKS build (scripts/build_variant.sh clocks -DFS_BLOCK_CLOCKS; FLINGSIM_LIB=variants/libfs_clocks.so), whose kernel
leaves its entry-to-exit shader clocks and the constant 100 MHz clock at entry / exit in row 95 of the episode's neighbour table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from flingbot_amd import sim as fsim

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = [int(a) for a in sys.argv[2:]] or [81, 120, 170]
ctx = fsim.FlingSim(n_envs=E, solver=2)
for e in range(E):
    bench.setup_episode(ctx.env(e), e)
done = 0
for f in frames:
    ctx.step(f - 1 - done); ctx.step(1); ctx.sync(); done = f
    clocks, start, end, cmean, cmax, cany = [], [], [], [], [], []
    for e in range(E):
        cnt, lists = ctx.get_last_neighbors(e)
        row = lists[:4, 95].astype(np.int64) & 0xffffffff
        clocks.append(int(row[0] | (row[1] << 32))); start.append(int(row[2])); end.append(int(row[3]))
        cmean.append(cnt.mean()); cmax.append(cnt.max()); cany.append(int((cnt > 0).sum()))
    clocks = np.array(clocks); start = np.array(start); end = np.array(end)
    us = ((end - start) & 0xffffffff) / 100.0
    order = np.argsort(-us)
    print("frame %d, %d episodes: launch %.1f us (first entry to last exit), starts within %.1f us" % (
        f, E, (((end - start.min()) & 0xffffffff).max()) / 100.0, ((start - start.min()) & 0xffffffff).max() / 100.0))
    print("   workgroup us: max %.1f p90 %.1f median %.1f p10 %.1f min %.1f; mean %.1f = %.3f of the longest; shader clock of the longest %.2f GHz, of the shortest %.2f GHz" % (
        us.max(), np.percentile(us, 90), np.median(us), np.percentile(us, 10), us.min(), us.mean(), us.mean() / us.max(),
        clocks[order[0]] / us[order[0]] / 1e3, clocks[order[-1]] / us[order[-1]] / 1e3))
    print("   longest:", ", ".join("%d: %.0f us (contacts mean %.2f max %d)" % (e, us[e], cmean[e], cmax[e]) for e in order[:8]))
    cany = np.array(cany)
    print("   particles with contacts per episode: max %d p90 %d median %d; episodes whose overflow (beyond the 1024 set slots) exceeds the queue's 1536: %d" % (
        cany.max(), np.percentile(cany, 90), np.median(cany), int((cany > 2560).sum())))
    print("   correlation of the duration with the mean contact count %.3f, with the longest list %.3f" % (
        np.corrcoef(us, cmean)[0, 1], np.corrcoef(us, cmax)[0, 1]))
