"""Development helper: SpatialValueNet forward (learning/nets.py:81-141 architecture) on the GPU, steady state:
the reference's module graph (MIOpen), BatchNorm folded + channels-last (MIOpen), and the hand-written forward
(fs_value_net_forward, csrc/fs_valuenet.hip)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flingbot_amd import nets

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nets.SpatialValueNet(rgb_only=True, device=dev).to(dev).eval()
flops_per_img = 306.7e6
for mode in ("reference module graph ", "BN folded, channels_last", "hand-written HIP        "):
    if mode.startswith("BN"):
        net.fold_batchnorm(hip=False)
    elif mode.startswith("hand"):
        net.fold_batchnorm(hip=True)
    for batch in ([int(a) for a in sys.argv[1:]] or [96, 768]):
        x = torch.rand(batch, 4, 64, 64, device=dev)
        with torch.no_grad():
            for _ in range(5):
                net(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                y = net(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("SpatialValueNet %s batch %4d: %.3f ms  (%.1f observations/s, %.2f TFLOP/s fp32)" % (
            mode, batch, dt * 1e3, batch / 96 / dt, batch * flops_per_img / dt / 1e12))
