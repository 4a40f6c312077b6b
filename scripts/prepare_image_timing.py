"""Development helper: prepare_image, host (scipy) vs device (fs_prepare_image), FlingBot sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flingbot_amd import nets

size, dim = 400, 64
rotations = [(2 * i / 11 - 1) * 90 for i in range(12)]
scales = [1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75]
tf = [(r, s) for r in rotations for s in scales]
g = torch.Generator().manual_seed(0)
img = torch.rand(4, size, size, generator=g)
t0 = time.perf_counter(); ref = nets.prepare_image(img, tf, dim); t_cpu = time.perf_counter() - t0
d = img.cuda()
out = nets.prepare_image(d, tf, dim); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = nets.prepare_image(d, tf, dim)
torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t0) / 20
diff = (out.cpu() - ref).abs()
print("prepare_image %d transforms of a 4x%dx%d observation -> %dx%d: host %.2f s, device %.3f ms (x%.0f); max |diff| %.3g, "
      "exactly equal %.4f %%" % (len(tf), size, size, dim, dim, t_cpu, t_gpu * 1e3, t_cpu / t_gpu, diff.max().item(),
                                100.0 * (diff == 0).float().mean().item()))
