"""Development helper: action selection at FlingBot size, host walk (numpy restatement of the reference) vs device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd.action import ActionSelector
from oracle import action as oa

rng = np.random.default_rng(0)
D, S, gd = 64, 400, 8
rotations = [(2 * i / 11 - 1) * 90 for i in range(12)]
scales = np.array([1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75])
yy, xx = np.mgrid[0:S, 0:S]
depth = np.full((S, S), 2.0, np.float32)
blob = ((xx - S * 0.45) ** 2 + (yy - S * 0.55) ** 2) < (S * 0.22) ** 2
depth[blob] = (1.97 - 0.04 * rng.random(blob.sum())).astype(np.float32)
values = rng.random((1, 96, D, D)).astype(np.float32)
cfg = dict(obs_dim=D, pix_grasp_dist=gd, pix_drag_dist=8, pix_place_dist=5, scales=scales, rotations=rotations, depth=depth,
           reach_distance_limit=0.6, stretchdrag_dist=0.3, grasp_height=0.02, left_arm_base=np.array([0.765, 0, 0]),
           right_arm_base=np.array([-0.765, 0, 0]))
t0 = time.perf_counter(); a, res, k = oa.get_max_value_valid_action(values, ["fling"], cfg); t_cpu = time.perf_counter() - t0
sel = ActionSelector(["fling"], rotations, D, gd, 8, 5, 0.6)
v = torch.tensor(values).cuda(); d = torch.tensor(depth).cuda()
sel.select(v, scales, d); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    a2, p2 = sel.select(v, scales, d)
torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t0) / 20
print("select_action over %d candidates: host walk %.1f ms (stopped at rank %d), device %.3f ms; same winner: %s" % (
    96 * (D - 2 * gd) ** 2, t_cpu * 1e3, k if k < 0 else int((np.sort(-values[:, :, gd:-gd, gd:-gd].ravel(), kind="stable") < -values[:, :, gd:-gd, gd:-gd].ravel()[k]).sum()),
    t_gpu * 1e3, a == a2 and p2["flat_index"] == k))
